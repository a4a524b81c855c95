#!/bin/bash
export MPHIP_ALLOW_ABLATED=1   # these variants are timing-only builds (csrc/mphip_ablate.h)
# same-box A/B of library variants on the dominant conv + the bench step.  usage: tools/ab_conv.sh out_dir lib1 lib2 ...
out=$1; shift
mkdir -p $out
for lib in "$@"; do
  for rep in 1 2; do
    MPHIP_LIB=$lib python tools/time_one_conv.py 96 96 16 64 64 3 8 1 >> $out/conv_ab.log 2>&1
  done
done
for lib in "$@"; do
  MPHIP_LIB=$lib python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], d['roofline']['launch_ms'])" >> $out/bench_ab.log
done
cat $out/conv_ab.log $out/bench_ab.log
