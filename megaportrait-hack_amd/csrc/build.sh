#!/bin/bash
# Builds libmphip.so (gfx950 only) next to this script's parent package directory.
# -ffp-contract=off: roundings are placed by hand (explicit fmaf where ATen fuses), see warp.hip.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="${1:-$here/../libmphip.so}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=default -Wall -Wno-unused-function ${MPHIP_EXTRA_FLAGS:-}"
objs=()
pids=()
bdir="${MPHIP_BUILD_DIR:-$here/build}"
mkdir -p "$bdir"
for f in api warp norm conv3d conv3d_f16x3 conv3d_f16x3_wino conv3d_f16x3_wino_pp conv3d_f16x3_wino_bt mfma_sol backward conv3d_bwd_f16x3 flowfield plan; do
  extra=""
  # (the role-split conv places its fp32 staging arithmetic by hand: no SLP packing into v_pk_*_f32, see the file's header)
  if [ "$f" == "conv3d_f16x3_wino_pp" ] || [ "$f" == "conv3d_f16x3_wino_bt" ]; then extra="-fno-slp-vectorize"; fi
  "$HIPCC" $FLAGS $extra -c "$here/$f.hip" -o "$bdir/$f.o" &
  pids+=($!)
  objs+=("$bdir/$f.o")
done
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$out" "${objs[@]}"
echo "built $out"
