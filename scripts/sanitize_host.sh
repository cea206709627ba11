#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the host-only part of the ingest (gtx_bed.cpp through gtx_packtool: line handling,
# tokenising, order checks, the packed-file writer): the CPU suite's packer tests against an instrumented build.  CPU only -- GPU
# sanitizers are not available on the pool.  usage: scripts/sanitize_host.sh [record-file]
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/ibm-cbc-genomic-tools_amd/csrc; T=$(mktemp -d)
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -I$R/include -pthread -o $T/gtx_packtool $C/gtx_packtool.cpp $C/gtx_bed.cpp -lz
cp $C/gtx_packtool $T/orig; cp $T/gtx_packtool $C/gtx_packtool
trap 'cp $T/orig $C/gtx_packtool; rm -rf $T' EXIT
out=$(cd $R && ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 python -m pytest tests/test_host_packer.py -x -q 2>&1 | tail -3)
echo "$out"
[ -n "$1" ] && { echo "g++ -fsanitize=address,undefined build of gtx_packtool (gtx_bed.cpp), tests/test_host_packer.py, $(date -u +%F):"; echo "$out"; } > "$1"
