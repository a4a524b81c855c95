#!/bin/bash
# SQ counters of K2's staged gather on the reference-like field (what bounds it: LDS conflicts? VALU? waits?).  usage: tools/pmc_k2.sh out_dir
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVES -d $out/a -- python tools/bench_warps.py 8 3 --only faithful > $out/a.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_WR SQ_LDS_ADDR_CONFLICT -d $out/b -- python tools/bench_warps.py 8 3 --only faithful > $out/b.log 2>&1
python - <<PY
import csv, glob, collections
for d in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % d, recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            for k in ("warp_gather_kernel", "warp_gather_dsum_kernel"):
                if k + "(" in row["Kernel_Name"] or k + "<" in row["Kernel_Name"]:
                    acc[(k, int(row["Dispatch_Id"]))][row["Counter_Name"]] += float(row["Counter_Value"])
    by = collections.defaultdict(list)
    for (k, disp), c in acc.items(): by[k].append(c)
    for k, lst in by.items():
        lst = lst[1:] or lst
        print(k, {n: round(sum(c[n] for c in lst) / len(lst)) for n in sorted(lst[0])})
PY
tail -3 $out/a.log $out/b.log
