#!/bin/bash
# Experiment builds of libw4a16.so: one translation unit recompiled with extra flags, the others taken from the
# shipped build's objects.   bash scripts/build_variant.sh <name> <unit> [flags...]
#   e.g. bash scripts/build_variant.sh stamps_wide w4a16_wide -DW4A16_STAMPS
# -> scripts/_exp/lib<name>.so (use it with W4A16_LIBRARY=scripts/_exp/lib<name>.so; delete when done: the
# directory travels to the GPU box).
set -e
name=${1:?name}; unit=${2:?unit}; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/vllm_quick_amd/csrc
mkdir -p "$R/scripts/_exp"
obj=$R/scripts/_exp/${name}_${unit}.o
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC "$@" -c -o "$obj" "$C/$unit.hip"
parts=()
for u in w4a16 w4a16_stock w4a16_streaming w4a16_decode w4a16_tiled w4a16_wide w4a16_wide_shared moe_route; do
  if [ "$u" = "$unit" ]; then parts+=("$obj"); else parts+=("$C/_obj/$u.o"); fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/scripts/_exp/lib${name}.so" "${parts[@]}"
rm -f "$obj"
echo "built scripts/_exp/lib${name}.so"
