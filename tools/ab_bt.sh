#!/bin/bash
# same-box A/B of the big-tile F(2,3) conv's timing ablations (csrc/mphip_ablate.h BT_ABL) on the dominant launch; interleaved rounds.
# usage: tools/ab_bt.sh out_file variant...   (variant = a BT_ABL mask built by tools/build_variant.sh bt_abl<mask>)
export MPHIP_ALLOW_ABLATED=1 MPHIP_WINOGRAD_MIN_TILES=1
out=$1; shift
: > $out
for rep in 1 2; do
  MPHIP_WINO_PP=1 timeout 120 python tools/time_one_conv.py 96 96 16 64 64 3 8 1 2>&1 | grep -v amdgpu.ids | sed 's/^/role-split /' >> $out
  MPHIP_WINO_PP=2 timeout 120 python tools/time_one_conv.py 96 96 16 64 64 3 8 1 2>&1 | grep -v amdgpu.ids | sed 's/^/big-tile   /' >> $out
  for v in "$@"; do
    MPHIP_WINO_PP=2 MPHIP_LIB=$PWD/build_variants/libmphip_bt_abl$v.so timeout 120 python tools/time_one_conv.py 96 96 16 64 64 3 8 1 2>&1 | grep -v amdgpu.ids | sed "s/^/abl $v /" >> $out
  done
done
cat $out
