#!/bin/bash
# Collects the round's evidence on a GPU box into gpurun_out/profiles_rNN/ — every JSON it writes carries the commit it was run at and the
# sha256 of the kernel sources it measured; bench.py quotes counters only when those stamps match the sources of the build.
# usage (from the build container):  gpurun -- "COMMIT=$(git rev-parse HEAD) ROUND=r06 bash tools/collect_profiles.sh"
cd $GRAFT_REPO_ROOT
R=${ROUND:-r06}
out=gpurun_out/profiles_$R; rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
stamp() {   # stamp <json file> <source files...>: adds _commit / _sources (sha256) to a JSON object in place
  python - "$@" <<'PY'
import hashlib, json, os, sys
path, srcs = sys.argv[1], sys.argv[2:]
rec = json.load(open(path))
rec["_commit"] = os.environ.get("COMMIT", "unknown")
rec["_sources"] = {s: hashlib.sha256(open(os.path.join("megaportrait-hack_amd", "csrc", s), "rb").read()).hexdigest() for s in srcs}
json.dump(rec, open(path, "w"), indent=1)
PY
}
# 1. kernel trace of the default bench loop (two batches in flight) -> per-kernel stats; and of the same steps on one stream -> timeline
rocprofv3 --kernel-trace -d $out/kt -- python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 --repeats 0 > $out/bench_under_trace.log 2>&1
db=$(find $out/kt -name "*.db" | head -1)
python tools/rocpd_summary.py $db $out/${R}_kernel_stats.csv
python tools/agg_summary.py $db 25 45 > $out/${R}_kernel_agg.txt
rm -rf $out/kt
rocprofv3 --kernel-trace -d $out/kt -- python tools/run_plan_steps.py 8 30 > /dev/null 2>&1
db=$(find $out/kt -name "*.db" | head -1)
python tools/lane_timeline.py $db 4.0 > $out/${R}_timeline_plan.txt
python tools/agg_summary.py $db 30 45 > $out/${R}_kernel_agg_one_stream.txt
rm -rf $out/kt
rocprofv3 --kernel-trace -d $out/kt -- python tools/run_plan_steps.py 1 30 > /dev/null 2>&1
db=$(find $out/kt -name "*.db" | head -1)
python tools/lane_timeline.py $db 1.6 > $out/${R}_timeline_plan_b1.txt
rm -rf $out/kt
# 1b. training step of config 3's shard (eager launches: 10 timed + 2 warm-up steps, every one the same kernels)
rocprofv3 --kernel-trace -d $out/kt -- python bench.py --mode train --batch 4 --steps 10 --warmup 2 > $out/train_under_trace.log 2>&1
db=$(find $out/kt -name "*.db" | head -1)
python tools/agg_summary.py $db 12 40 > $out/${R}_train_kernel_agg.txt
rm -rf $out/kt
# 2. the same line without the profiler, same box
python bench.py --no-cpu-baseline --extras-budget 30 --steps 20 --warmup 5 2>/dev/null | tail -1 > $out/${R}_bench_line_same_box_as_trace.json
# 3. SQ / HBM counters of the dominant conv (the F(2,3) kernel; separate passes, MI355X_MICROARCH.md)
bash tools/pmc_conv.sh $out/pmc_conv conv3d_k3_f16x3_wino_pp_kernel > $out/pmc_conv.log 2>&1
cp $out/pmc_conv/conv_pmc.json $out/${R}_pmc_conv.json && stamp $out/${R}_pmc_conv.json conv3d_f16x3_wino_pp.hip mphip_wino_tile.h mphip_f16x3.h
rm -rf $out/pmc_conv
# 4. HBM counters of K2 / K3 (B = 8)
bash tools/pmc_warps.sh $out/pmc_warps 8 > $out/pmc_warps.log 2>&1
python tools/pmc_warps_profile.py $out/pmc_warps/summary.json $out/${R}_pmc_warps.json > /dev/null && stamp $out/${R}_pmc_warps.json warp.hip
rm -rf $out/pmc_warps
# 4b. the three F(2,3) schedules side by side (SQ / LDS counters), the two-frame mode at G3d's 2x8x8 level, the big-tile kernel's check
bash tools/pmc_bt.sh $out/pmc_bt > $out/${R}_pmc_schedules.txt 2>&1; rm -rf $out/pmc_bt
python tools/d2_check.py 2>&1 | grep -v amdgpu > $out/${R}_two_frame_mode.txt
python tools/bt_check.py 2>&1 | grep -v amdgpu > $out/${R}_big_tile_check.txt
# 5. batch sweep through the plan, one generator alone
python tools/bench_plan.py 2>&1 | grep -v amdgpu > $out/${R}_plan_vs_perop.txt
python tools/bench_generator.py 2>&1 | grep -v amdgpu > $out/${R}_generator.txt
ls -la $out
