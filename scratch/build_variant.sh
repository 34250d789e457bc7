#!/bin/bash
# usage: scratch/build_variant.sh <name> <file.hip | all> [-DFLAG=V ...]   -> scratch/variants/lib_<name>.so (the product library with ONE
# translation unit -- or, with "all", every one -- recompiled under extra defines; select it with CRFCONV_LIB=... for A/B runs on the GPU box)
set -e
cd "$(dirname "$0")/../crfconv_amd/csrc"
name=$1; shift
src=$1; shift
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-pass-failed"
if [ "$src" = all ]; then
  tmpd=../../scratch/variants/build_$name
  mkdir -p $tmpd
  for f in *.hip; do echo "$f"; done | xargs -P 8 -I{} sh -c "$CC $* -c {} -o $tmpd/\$(basename {} .hip).o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/variants/lib_$name.so $tmpd/*.o
  rm -rf $tmpd
  exit 0
fi
obj=build/${src%.hip}.o
tmp=../../scratch/variants/${src%.hip}_$name.o
$CC "$@" -c $src -o $tmp
objs=$(ls build/*.o | grep -v "^$obj$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/variants/lib_$name.so $objs $tmp
rm -f $tmp
