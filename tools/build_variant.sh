#!/bin/bash
# builds a library variant that recompiles ONE translation unit with extra flags and links it with the main build's other objects.
# usage: tools/build_variant.sh <name> <unit> "<extra flags>"   ->  build_variants/libmphip_<name>.so
set -euo pipefail
name=$1; unit=$2; flags=$3
here="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
src=$here/megaportrait-hack_amd/csrc
mkdir -p $here/build_variants /tmp/bv_$name
extra=""
if [ "$unit" == "conv3d_f16x3_wino_pp" ] || [ "$unit" == "conv3d_f16x3_wino_bt" ]; then extra="-fno-slp-vectorize"; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=default -Wall -Wno-unused-function $extra $flags -c $src/$unit.hip -o /tmp/bv_$name/$unit.o
objs=()
for f in api warp norm conv3d conv3d_f16x3 conv3d_f16x3_wino conv3d_f16x3_wino_pp conv3d_f16x3_wino_bt mfma_sol backward conv3d_bwd_f16x3 flowfield plan; do
  if [ "$f" == "$unit" ]; then objs+=(/tmp/bv_$name/$unit.o); else objs+=($src/build/$f.o); fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $here/build_variants/libmphip_$name.so "${objs[@]}"
echo built build_variants/libmphip_$name.so
