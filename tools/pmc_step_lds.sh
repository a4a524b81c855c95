#!/bin/bash
# LDS-pipe counters of every kernel of a plan step (B = 8): bank-conflict cycles as a share of the pipe's active cycles, and the pipe's
# share of the launch (r04: K2's tail was its tap reads' bank conflicts).  usage: tools/pmc_step_lds.sh out_dir
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $out/lds -- python tools/run_plan_steps.py 8 6 > $out/lds.log 2>&1
python - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); meta = {}
for f in glob.glob("$out/lds/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        k = (row["Kernel_Name"], int(row["Dispatch_Id"]))
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        meta[k] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"]), row.get("Grid_Size", ""))
agg = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0, 0])
for (name, disp), c in acc.items():
    n = re.sub(r"^mphip::", "", re.sub(r"^void ", "", re.sub(r"\(.*", "", name)))
    a = agg[(n[:56], meta[(name, disp)][1])]
    a[0] += c.get("SQ_LDS_IDX_ACTIVE", 0); a[1] += c.get("SQ_LDS_BANK_CONFLICT", 0); a[2] += c.get("SQ_INSTS_LDS", 0); a[3] += 1; a[4] += meta[(name, disp)][0]
print("%-56s %-9s %8s %10s %9s %8s" % ("kernel", "grid", "us", "LDS act k", "conflict", "LDS busy"))
for (n, grid), a in sorted(agg.items(), key=lambda kv: -kv[1][4])[:40]:
    us = a[4] / a[3] / 1e3
    act = a[0] / a[3]
    busy = act / 256 / (us * 2.2e3) if us else 0   # LDS-active cycles per CU / launch cycles at ~2.2 GHz
    print("%-56s %-9s %8.1f %10.0f %8.1f%% %7.1f%%  x%d" % (n, grid, us, act / 1e3, 100 * a[1] / max(1.0, a[0]), 100 * busy, a[3]))
PY
