#!/bin/bash
# SQ / LDS counters of the two F(2,3) schedules on the dominant launch, side by side (separate passes per counter set, per kernel).
# usage (GPU box): tools/pmc_bt.sh out_dir
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MPHIP_WINOGRAD_MIN_TILES=1
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $out/sq_counters_avail.txt
for mode in 1 2; do
  MPHIP_WINO_PP=$mode rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAVES -d $out/a$mode -- python tools/run_one_conv.py 96 96 16 64 64 3 8 4 > $out/a$mode.log 2>&1
  MPHIP_WINO_PP=$mode rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES -d $out/b$mode -- python tools/run_one_conv.py 96 96 16 64 64 3 8 4 > $out/b$mode.log 2>&1
  MPHIP_WINO_PP=$mode rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES -d $out/c$mode -- python tools/run_one_conv.py 96 96 16 64 64 3 8 4 > $out/c$mode.log 2>&1
done
python - <<PY
import csv, glob, collections, json
res = {}
for mode, name in ((1, "role-split"), (2, "big-tile")):
    tot = {}
    for s in "abc":
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for f in glob.glob("$out/%s%d/**/*counter_collection.csv" % (s, mode), recursive=True):
            for row in csv.DictReader(open(f, newline="")):
                if "wino_pp_kernel" in row["Kernel_Name"] or "wino_bt_kernel" in row["Kernel_Name"]:
                    acc[int(row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
        ids = sorted(acc)[1:]
        for n in sorted({k for i in ids for k in acc[i]}):
            tot.setdefault(n, sum(acc[i][n] for i in ids) / max(1, len(ids)))
    res[name] = tot
json.dump(res, open("$out/pmc_bt.json", "w"), indent=1)
names = sorted(set(res["role-split"]) | set(res["big-tile"]))
for n in names:
    print("%-28s %16.0f %16.0f" % (n, res["role-split"].get(n, float("nan")), res["big-tile"].get(n, float("nan"))))
PY
