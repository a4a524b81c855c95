#!/bin/bash
# LDS bank-conflict counters of the role-split F(2,3) conv on the dominant launch for a list of library variants (which LDS operations conflict?)
# usage (GPU box): tools/pmc_lds_variants.sh out_dir lib...
out=$1; shift; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MPHIP_WINOGRAD_MIN_TILES=1 MPHIP_ALLOW_ABLATED=1 MPHIP_WINO_PP=1
i=0
for lib in "$@"; do
  i=$((i+1))
  MPHIP_LIB=$PWD/$lib rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_LDS_ADDR_CONFLICT SQ_WAVES -d $out/v$i -- python tools/run_one_conv.py 96 96 16 64 64 3 8 4 > $out/v$i.log 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$out/v$i/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        if "wino_pp_kernel" in row["Kernel_Name"] or "wino_bt_kernel" in row["Kernel_Name"]:
            acc[int(row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
ids = sorted(acc)[1:]
m = {n: sum(acc[i][n] for i in ids) / max(1, len(ids)) for n in sorted({k for i in ids for k in acc[i]})}
print("$lib:", {k: round(v / 1e6, 2) for k, v in m.items()})
PY
  rm -rf $out/v$i
done
