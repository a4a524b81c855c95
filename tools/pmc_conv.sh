#!/bin/bash
# SQ counters of the dominant f16x3 conv (one pass; MI355X_MICROARCH.md PMC slots: 8 SQ counters) + its HBM traffic passes.
# usage (GPU box): tools/pmc_conv.sh out_dir
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES -d $out/sq -- python tools/run_one_conv.py 96 96 16 64 64 3 8 4 > $out/sq.log 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d $out/$ctr -- python tools/run_one_conv.py 96 96 16 64 64 3 8 4 > $out/$ctr.log 2>&1
done
python - <<PY
import csv, glob, json, collections
res = {}
for d in ("sq", "FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % d, recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            if "conv3d_k3_f16x3_kernel" in row["Kernel_Name"]:
                acc[int(row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
    ids = sorted(acc)[1:]
    if ids:
        names = sorted({k for i in ids for k in acc[i]})
        for n in names:
            res[n] = sum(acc[i][n] for i in ids) / len(ids)
json.dump(res, open("$out/conv_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
