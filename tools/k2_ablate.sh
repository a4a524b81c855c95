#!/bin/bash
export MPHIP_ALLOW_ABLATED=1   # these variants are timing-only builds (csrc/mphip_ablate.h)
# same-box A/B of library variants on K2 / K3 alone, per-kernel times from a kernel trace.  usage: tools/k2_ablate.sh lib1 lib2 ...
export TMPDIR=/tmp
for lib in "$@"; do
  rm -rf /tmp/kt; MPHIP_LIB=$PWD/$lib rocprofv3 --kernel-trace -d /tmp/kt -- python tools/bench_warps.py 8 20 --only faithful > /dev/null 2>&1
  db=$(find /tmp/kt -name "*.db" | head -1); echo $lib; python tools/agg_summary.py $db 23 12 | grep -i "warp\|total" | head -8
  MPHIP_LIB=$PWD/$lib python tools/bench_warps.py 8 30 2>&1 | grep "K2"
done
