#!/bin/bash
# HBM-side traffic of EVERY kernel of a plan step (B = 8): FETCH_SIZE (x2: MI355X_MICROARCH.md) and WRITE_SIZE per launch, with the launch time —
# a kernel that moves far more than its tensors is over-fetching (r04: FlowField's level-1 blocks read 113 MB for 12.6 MB of weights).
# usage: tools/pmc_step_traffic.sh out_dir
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for ctr in ${CTRS:-FETCH_SIZE WRITE_SIZE}; do
  rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d $out/$ctr -- python tools/run_plan_steps.py 8 6 > $out/$ctr.log 2>&1
done
python - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": 0, "ns": 0})
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(float); meta = {}
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % ctr, recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            if row["Counter_Name"] != ctr: continue
            k = (row["Kernel_Name"], int(row["Dispatch_Id"]))
            per[k] += float(row["Counter_Value"])
            meta[k] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"]), row.get("Grid_Size", ""))
    for (name, disp), v in per.items():
        n = re.sub(r"^void ", "", re.sub(r"\(.*", "", name))
        n = re.sub(r"^mphip::", "", n)
        key = (n[:58], meta[(name, disp)][1])
        acc[key][ctr] += v * 1024
        if ctr == "FETCH_SIZE":
            acc[key]["n"] += 1; acc[key]["ns"] += meta[(name, disp)][0]
rows = []
for (n, grid), a in acc.items():
    if not a["n"]: continue
    rd, wr, us = 2 * a["FETCH_SIZE"] / a["n"], a["WRITE_SIZE"] / a["n"], a["ns"] / a["n"] / 1e3
    rows.append((us * a["n"], n, grid, rd, wr, us, a["n"]))
print("%-58s %-10s %9s %9s %8s %7s" % ("kernel", "grid", "read MB", "write MB", "us", "TB/s"))
for tot, n, grid, rd, wr, us, cnt in sorted(rows, reverse=True)[:60]:
    print("%-58s %-10s %9.2f %9.2f %8.1f %7.2f   x%d" % (n, grid, rd / 1e6, wr / 1e6, us, (rd + wr) / us / 1e6, cnt))
PY
