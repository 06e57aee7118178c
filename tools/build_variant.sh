#!/bin/bash
# Development aid: build a variant of the kernel library from an alternative gemm_conv.hip (the slow file) next to the production one:
#   tools/build_variant.sh <name> <path/to/gemm_conv_variant.hip>  ->  layoutdetr_amd/lib/variants/libldetr_hip_<name>.so
# Select it at run time with LDETR_LIB=<that path> (layoutdetr_amd/_lib.py).  The other objects are the production build's.
set -e
name=$1; src=$2
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/layoutdetr_amd/lib/variants; mkdir -p $out /tmp/ldetr_var_$name
cp $src $root/layoutdetr_amd/csrc/.variant_$name.hip
hipcc -DLDETR_TILE_TRACE=0 -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fno-gpu-rdc -Wno-unused-result \
  -Rpass-analysis=kernel-resource-usage -x hip -c $root/layoutdetr_amd/csrc/.variant_$name.hip -o /tmp/ldetr_var_$name/gemm_conv.o 2> $out/$name.remarks.txt
rm -f $root/layoutdetr_amd/csrc/.variant_$name.hip
objs=$(ls $root/layoutdetr_amd/lib/obj/*.o | grep -v gemm_conv)
hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc -o $out/libldetr_hip_$name.so $objs /tmp/ldetr_var_$name/gemm_conv.o
echo built $out/libldetr_hip_$name.so
