#!/bin/bash
# Development aid: build a variant of the kernel library with one source file compiled differently (other source / extra -D flags), next to the
# production one:  tools/build_variant.sh <name> <source.hip in csrc/ or a path> [extra hipcc flags...]
#   -> layoutdetr_amd/lib/variants/libldetr_hip_<name>.so ; select it at run time with LDETR_LIB=<that path> (layoutdetr_amd/_lib.py).
# The other objects are the production build's (run `python -m layoutdetr_amd.build` first).
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
[ -f "$src" ] || src=$root/layoutdetr_amd/csrc/$src
base=$(basename $src)
out=$root/layoutdetr_amd/lib/variants; mkdir -p $out /tmp/ldetr_var_$name
hipcc -DLDETR_TILE_TRACE=0 -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fno-gpu-rdc -Wno-unused-result "$@" \
  -I$root/layoutdetr_amd/csrc -Rpass-analysis=kernel-resource-usage -x hip -c $src -o /tmp/ldetr_var_$name/$base.o 2> $out/$name.remarks.txt
objs=$(ls $root/layoutdetr_amd/lib/obj/*.o | grep -v "/$base.o")
hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc -o $out/libldetr_hip_$name.so $objs /tmp/ldetr_var_$name/$base.o
echo built $out/libldetr_hip_$name.so
