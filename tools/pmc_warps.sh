#!/bin/bash
# HBM counters of K2/K3 (MI355X_MICROARCH.md: separate --pmc passes; FETCH_SIZE x2 on gfx950 for wide coalesced reads).
# usage (on the GPU box): tools/pmc_warps.sh out_dir [batches, default "8 16"]
out=$1; mkdir -p $out
BATCHES=${2:-8 16}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for B in $BATCHES; do
  for kind in faithful smooth; do
    for ctr in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d $out/pmc_${kind}_B${B}_$ctr -- python tools/bench_warps.py $B 3 --only $kind > $out/pmc_${kind}_B${B}_$ctr.log 2>&1
    done
  done
done
python tools/pmc_summary.py $out > $out/summary.json
cat $out/summary.json
