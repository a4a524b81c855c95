#!/bin/bash
# same-box A/B of the role-split F(2,3) conv's timing ablations (csrc/mphip_ablate.h PP_ABL) on the dominant launch; interleaved rounds.
# usage: tools/ab_pp.sh out_file variant...   (variant = a library built by tools/build_variant.sh <variant> conv3d_f16x3_wino_pp ...)
export MPHIP_ALLOW_ABLATED=1 MPHIP_WINOGRAD_MIN_TILES=1 MPHIP_WINO_PP=1
out=$1; shift
: > $out
for rep in 1 2 3; do
  timeout 120 python tools/time_one_conv.py 96 96 16 64 64 3 8 1 2>&1 | grep -v amdgpu.ids | sed 's/^/product /' >> $out
  for v in "$@"; do
    MPHIP_LIB=$PWD/build_variants/libmphip_$v.so timeout 120 python tools/time_one_conv.py 96 96 16 64 64 3 8 1 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /" >> $out
  done
done
awk '{print $1, $(NF-3), $(NF-2)}' $out
