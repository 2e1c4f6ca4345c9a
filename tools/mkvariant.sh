#!/bin/bash
# Build an A/B variant of the library: recompile ONE kernel file with extra flags and link it with
# the objects of the regular build.  Usage: tools/mkvariant.sh NAME FILE.hip [-DFLAG ...]
# -> variants/lib_NAME.so (select with MUYGPYS_HIP_LIB=variants/lib_NAME.so; git-ignored)
set -e
name=$1; file=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/muygpys_amd/csrc/$file
obj=$root/variants/${file%.hip}_$name.o
common="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-slp-vectorize -mllvm -pragma-unroll-threshold=1000000 -Wno-pass-failed"
/opt/rocm/bin/hipcc $common "$@" -c "$src" -o "$obj"
others=$(ls $root/muygpys_amd/build/*.o | grep -v "/${file%.hip}.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/variants/lib_$name.so $obj $others
echo built variants/lib_$name.so
