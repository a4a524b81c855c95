#!/bin/bash
# L1 -> L2 read / write requests of every kernel of a plan step (B = 8): a kernel whose L2 traffic is a multiple of its tensors over-fetches
# lines (r04: FlowField level 1 touched every 128-byte line of its weights for 12 of 108 bytes).  usage: tools/pmc_step_l2.sh out_dir
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum -d $out/l2 -- python tools/run_plan_steps.py 8 6 > $out/l2.log 2>&1
python - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); meta = {}
for f in glob.glob("$out/l2/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        k = (row["Kernel_Name"], int(row["Dispatch_Id"]))
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        meta[k] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"]), row.get("Grid_Size", ""))
agg = collections.defaultdict(lambda: [0.0, 0.0, 0, 0])
for (name, disp), c in acc.items():
    n = re.sub(r"^mphip::", "", re.sub(r"^void ", "", re.sub(r"\(.*", "", name)))
    a = agg[(n[:58], meta[(name, disp)][1])]
    a[0] += c.get("TCP_TCC_READ_REQ_sum", 0); a[1] += c.get("TCP_TCC_WRITE_REQ_sum", 0); a[2] += 1; a[3] += meta[(name, disp)][0]
print("%-58s %-9s %10s %10s %8s %9s" % ("kernel", "grid", "rd req k", "wr req k", "us", "req/us"))
for (n, grid), a in sorted(agg.items(), key=lambda kv: -kv[1][3])[:45]:
    print("%-58s %-9s %10.1f %10.1f %8.1f %9.0f  x%d" % (n, grid, a[0] / a[2] / 1e3, a[1] / a[2] / 1e3, a[3] / a[2] / 1e3, (a[0] + a[1]) / a[3] * 1e3, a[2]))
PY
