"""Summarises rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/pmc_warps.sh into one JSON: per (field kind, batch,
kernel) the mean counter value per dispatch, in bytes.  FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE
counts wide coalesced reads at half their bytes (MI355X_MICROARCH.md, HBM section; re-calibrated r01: tools/pmc_calibrate.py),
so reads are also given x2.  usage: python tools/pmc_summary.py <dir>"""
import csv, glob, json, os, re, sys
from collections import defaultdict

root = sys.argv[1]
KEEP = ("warp_gather_kernel", "warp_gather_columns_kernel", "warp_gather_direct_kernel", "warp_gather_dsum_kernel", "warp_coords_kernel", "warp_corner_image_kernel")
out = {}
for d in sorted(glob.glob(os.path.join(root, "pmc_*_B*_*"))):
    if not os.path.isdir(d):
        continue
    m = re.match(r"pmc_(\w+?)_B(\d+)_(FETCH_SIZE|WRITE_SIZE)$", os.path.basename(d))
    if not m:
        continue
    kind, B, ctr = m.group(1), int(m.group(2)), m.group(3)
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"]
                for k in KEEP:
                    if k + "(" in name or k + "<" in name:
                        if row["Counter_Name"] == ctr:
                            acc[k].append((int(row["Dispatch_Id"]), float(row["Counter_Value"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    for k, vals in acc.items():
        # a dispatch reports one row per XCD/instance: sum per dispatch, then average over dispatches (skip the first = warm-up)
        per = defaultdict(float)
        dur = {}
        for disp, v, ns in vals:
            per[disp] += v
            dur[disp] = ns
        ids = sorted(per)[1:] or sorted(per)
        mean_kb = sum(per[i] for i in ids) / len(ids)
        key = f"{k} / {kind} / B={B}"
        rec = out.setdefault(key, {})
        rec[ctr + "_bytes"] = round(mean_kb * 1024)
        rec["dispatches"] = len(ids)
        rec["mean_us_under_pmc"] = round(sum(dur[i] for i in ids) / len(ids) / 1e3, 1)
for key, rec in out.items():
    if "FETCH_SIZE_bytes" in rec:
        rec["read_bytes_x2_calibrated"] = 2 * rec["FETCH_SIZE_bytes"]
    if "FETCH_SIZE_bytes" in rec and "WRITE_SIZE_bytes" in rec:
        rec["traffic_bytes_per_launch"] = rec["read_bytes_x2_calibrated"] + rec["WRITE_SIZE_bytes"]
print(json.dumps(out, indent=1, sort_keys=True))
