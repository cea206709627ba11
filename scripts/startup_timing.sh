#!/bin/bash
# process-level cost of touching the GPU at all: a program that only initialises HIP (and optionally allocates MB)
cd "$GRAFT_REPO_ROOT"
t() { s=$(date +%s%N); "$@" > /dev/null 2>&1; e=$(date +%s%N); echo "$(( (e - s) / 1000000 )) ms : $*"; }
t ./scripts/hipmin.bin
t ./scripts/hipmin.bin
t ./scripts/hipmin.bin 0 quick
t ./scripts/hipmin.bin 1024
t ./scripts/hipmin.bin 1024 quick
t ./scripts/hipmin.bin 4096
