#!/bin/bash
# variant build of libgtx.so for A/B experiments: scripts/x_build.sh NAME "-DFLAG ..." -> ab/libgtx_NAME.so (ab/ is not tracked)
set -e
name=$1; flags=$2
src=ibm-cbc-genomic-tools_amd/csrc
mkdir -p ab/obj_$name
for f in gtx_kernels gtx_bucket gtx_special gtx_scanown gtx_capi gtx_group gtx_text gtx_pairs gtx_perm; do
  extra=""; [ $f = gtx_perm ] && extra="-ffp-contract=off"
  if [ ! -f ab/obj_$name/$f.o ] || [ $src/$f.hip -nt ab/obj_$name/$f.o ] || [ -n "$X_FORCE" ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -I$src -Wno-unused-result $extra $flags -c $src/$f.hip -o ab/obj_$name/$f.o &
  fi
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ab/libgtx_$name.so ab/obj_$name/*.o -ldl
echo built ab/libgtx_$name.so
