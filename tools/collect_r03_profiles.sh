#!/bin/bash
# Collects the round-3 evidence on a GPU box into gpurun_out/r03_profiles/ (copied into profiles/ afterwards).
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_profiles; rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
# 1. kernel trace of the default bench loop (two batches in flight, C-side plans, demand-driven tail): 20 + 5 steps
rocprofv3 --kernel-trace -d $out/kt -- python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > $out/bench_under_trace.log 2>&1
db=$(find $out/kt -name "*.db" | head -1)
python tools/rocpd_summary.py $db $out/r03_kernel_stats.csv
python tools/agg_summary.py $db 25 45 > $out/r03_kernel_agg.txt
python tools/lane_timeline.py $db 8.0 > $out/r03_timeline_two_in_flight.txt
rm -rf $out/kt
# 1b. the same steps on one stream (the `one_in_flight` leg's loop): timeline of one step
rocprofv3 --kernel-trace -d $out/kt -- python tools/run_plan_steps.py 8 30 > /dev/null 2>&1
db=$(find $out/kt -name "*.db" | head -1)
python tools/lane_timeline.py $db 4.6 > $out/r03_timeline_plan.txt
rm -rf $out/kt
rocprofv3 --kernel-trace -d $out/kt -- python tools/run_plan_steps.py 1 30 > /dev/null 2>&1
db=$(find $out/kt -name "*.db" | head -1)
python tools/lane_timeline.py $db 1.8 > $out/r03_timeline_plan_b1.txt
rm -rf $out/kt
# 2. the same line without the profiler (same box), and its serial / full-tail legs
python bench.py --no-cpu-baseline --extras-budget 1 --steps 40 --warmup 5 2>/dev/null | tail -1 > $out/r03_bench_line_same_box_as_trace.json
# 3. SQ / HBM counters of the dominant conv (separate passes)
bash tools/pmc_conv.sh $out/pmc_conv > $out/pmc_conv.log 2>&1
cp $out/pmc_conv/conv_pmc.json $out/r03_pmc_conv_raw.json
rm -rf $out/pmc_conv
# 4. batch sweep through the plan + isolated latencies, generator alone, planner sweep, k=1 shortcuts
python tools/bench_plan.py 2>&1 | grep -v amdgpu > $out/r03_plan_vs_perop.txt
python tools/bench_generator.py 2>&1 | grep -v amdgpu > $out/r03_generator.txt
python tools/sweep_conv_plans.py 8 1 2>&1 | grep -v amdgpu > $out/r03_conv_plan_sweep.txt
MPHIP_F16X3_K1_MIN=64 python tools/time_k1.py 8 1 2>&1 | grep -v amdgpu > $out/r03_k1_shortcuts.txt
ls -la $out
