#!/bin/bash
# SQ counters of the dominant f16x3 conv (one pass; MI355X_MICROARCH.md PMC slots: 8 SQ counters) + its HBM traffic passes.
# usage (GPU box): tools/pmc_conv.sh out_dir [kernel name substring, default: the F(2,3) kernel]
out=$1; mkdir -p $out
KERNEL=${2:-conv3d_k3_f16x3_wino_pp_kernel}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES -d $out/sq -- python tools/run_one_conv.py 96 96 16 64 64 3 8 4 > $out/sq.log 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d $out/$ctr -- python tools/run_one_conv.py 96 96 16 64 64 3 8 4 > $out/$ctr.log 2>&1
done
python - <<PY
import csv, glob, json, collections
res = {}
KERNEL = "$KERNEL"
for d in ("sq", "FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % d, recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            if KERNEL in row["Kernel_Name"]:
                acc[int(row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
    ids = sorted(acc)[1:]
    if ids:
        names = sorted({k for i in ids for k in acc[i]})
        for n in names:
            res[n] = sum(acc[i][n] for i in ids) / len(ids)
# derived figures (MI355X_MICROARCH.md: SQ_* wave counters are quad-cycles except SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES; FETCH_SIZE / WRITE_SIZE in KB,
# FETCH_SIZE counts wide coalesced reads at half their bytes on gfx950: x2, calibrated r01 with tools/pmc_calibrate.py)
der = {}
if "SQ_WAVE_CYCLES" in res and res.get("SQ_WAVES"):
    wc = res["SQ_WAVE_CYCLES"]
    der["waves"] = res["SQ_WAVES"]
    der["wave_time_split"] = {"issuing (ACTIVE_INST_ANY)": round(res["SQ_ACTIVE_INST_ANY"] / wc, 3), "parked on s_waitcnt/barrier (WAIT_ANY)": round(res["SQ_WAIT_ANY"] / wc, 3),
                              "issue-stalled (WAIT_INST_ANY)": round(res["SQ_WAIT_INST_ANY"] / wc, 3)}
    der["mfma_busy_cycles_per_simd"] = res["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0
    der["cycles_per_wave"] = 4.0 * wc / res["SQ_WAVES"]
    der["mfma_pipe_busy_frac"] = round(der["mfma_busy_cycles_per_simd"] / (der["cycles_per_wave"]), 3)
if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
    der["fetch_bytes_x2_calibrated"] = round(2 * 1024 * res["FETCH_SIZE"])
    der["write_bytes"] = round(1024 * res["WRITE_SIZE"])
    der["traffic_bytes_per_launch"] = der["fetch_bytes_x2_calibrated"] + der["write_bytes"]
    der["algorithmic_bytes_per_launch"] = 2 * 8 * 96 * 16 * 64 * 64 * 4
rec = {"_method": "rocprofv3 --kernel-trace --output-format csv --pmc <8 SQ counters, one pass> / --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tools/pmc_conv.sh -> "
                  "tools/run_one_conv.py 96 96 16 64 64 3 8 4 (the dominant conv, B=8, standalone back-to-back full launches); rows summed per dispatch, mean over dispatches 2..4",
       "kernel": KERNEL + " 96->96 @16x64x64 B=8", "counters": res, "derived": der}
json.dump(rec, open("$out/conv_pmc.json", "w"), indent=1)
print(json.dumps(rec, indent=1))
PY
