#!/bin/bash
# LDS-pipe counters of the conv kernels: bank conflicts as a share of the LDS pipe's active cycles.  usage: tools/pmc_lds_conv.sh out_dir
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for shape in "96 96 16 64 64 3 8" "192 192 8 32 32 3 8" "384 384 4 16 16 3 8" "768 768 2 8 8 3 8" "96 192 8 32 32 1 8"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_LDS_ADDR_CONFLICT SQ_WAVES -d $out/s$i -- python tools/run_one_conv.py $shape 4 > $out/s$i.log 2>&1
done
python - <<PY
import csv, glob, collections
for i in range(1, 6):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob("$out/s%d/**/*counter_collection.csv" % i, recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            n = row["Kernel_Name"]
            if "conv3d" in n and "pack" not in n:
                acc[(n[:60], int(row["Dispatch_Id"]))][row["Counter_Name"]] += float(row["Counter_Value"])
    by = collections.defaultdict(list)
    for (k, disp), c in sorted(acc.items()): by[k].append(c)
    for k, lst in by.items():
        lst = lst[1:] or lst
        m = {n: sum(c[n] for c in lst) / len(lst) for n in lst[0]}
        print("shape %d %s: LDS active %.1f M  bank-conflict %.1f M (%.1f %%)  LDS insts %.2f M  wave cycles(quad) %.1f M" % (
            i, k, m.get("SQ_LDS_IDX_ACTIVE", 0) / 1e6, m.get("SQ_LDS_BANK_CONFLICT", 0) / 1e6,
            100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, m.get("SQ_LDS_IDX_ACTIVE", 1)), m.get("SQ_INSTS_LDS", 0) / 1e6, m.get("SQ_WAVE_CYCLES", 0) / 1e6))
PY
