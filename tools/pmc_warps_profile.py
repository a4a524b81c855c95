"""Wrap the summary of tools/pmc_warps.sh (tools/pmc_summary.py output) into profiles/r02_pmc_warps.json: the per-kernel counters under
"kernels" (what bench.py's roofline_hbm reads) plus the method and a one-line reading per (kernel family, field kind) at B=8.
usage: python tools/pmc_warps_profile.py <summary.json> <out.json>"""
import json
import sys

k = json.load(open(sys.argv[1]))
FAM = {"K2": ["warp_coords_kernel", "warp_corner_image_kernel", "warp_gather_kernel", "warp_gather_columns_kernel", "warp_gather_direct_kernel"],
       "K3": ["warp_coords_kernel", "warp_gather_dsum_kernel"]}
ALG = {"K2": 428.0, "K3": 239.0}  # MB at B=8 (53.5 / 29.9 MB per frame)
reading = {}
for fam, names in FAM.items():
    for kind in ("faithful", "smooth"):
        recs = [(n, k.get(f"{n} / {kind} / B=8")) for n in names]
        recs = [(n, r) for n, r in recs if r]
        if not recs:
            continue
        mb = sum(r["traffic_bytes_per_launch"] for _, r in recs) / 1e6
        us = sum(r["mean_us_under_pmc"] for _, r in recs)
        reading[f"{fam} {kind} B=8"] = (f"{mb:.0f} MB of counted traffic per call ({mb / ALG[fam]:.2f}x the {ALG[fam]:.0f} MB algorithmic) in {us:.0f} us of "
                                        f"kernel time under the profiler = {mb / us:.2f} TB/s counted; kernels: "
                                        + ", ".join(f"{n.replace('warp_', '').replace('_kernel', '')} {r['mean_us_under_pmc']:.0f} us" for n, r in recs))
out = {"_method": "rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE / --pmc WRITE_SIZE in SEPARATE passes (tools/pmc_warps.sh -> "
                  "tools/bench_warps.py B iters --only kind), counter rows summed per dispatch, mean over dispatches 2..N, KB*1024. FETCH_SIZE counts "
                  "wide coalesced reads at half their bytes on gfx950 (MI355X_MICROARCH.md HBM section; calibrated in r01 on known byte counts, "
                  "tools/pmc_calibrate.py) -> read_bytes_x2_calibrated; WRITE_SIZE is exact. The counters sit on the L2's fabric side: Infinity-Cache "
                  "hits are included. Field kinds: faithful = what the reference's generators produce (samples in the low corner), smooth = "
                  "pixel-space identity + smooth +-3 voxel displacement (a field that travels through the volume). Volume 96x16x64x64 fp32: 25.17 MB "
                  "per frame; algorithmic bytes per frame (SURVEY.md 8d): K2 53.5 MB, K3 29.9 MB.",
       "_reading": reading, "kernels": k}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(reading, indent=1))
