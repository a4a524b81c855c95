#!/bin/bash
# Collects the round-3 evidence on a GPU box into gpurun_out/r03_profiles/ (copied into profiles/ afterwards).
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_profiles; mkdir -p $out
export TMPDIR=/tmp
# 1. kernel trace of the default bench loop (C-side plan, demand-driven tail)
rocprofv3 --kernel-trace -d $out/kt -- python bench.py --no-extras --no-cpu-baseline --steps 25 --warmup 5 > $out/bench_under_trace.log 2>&1
db=$(find $out/kt -name "*.db" | head -1)
python tools/rocpd_summary.py $db $out/r03_kernel_stats.csv
python tools/agg_summary.py $db 30 40 > $out/r03_kernel_agg.txt
python tools/lane_timeline.py $db 4.6 > $out/r03_timeline_plan.txt
rm -rf $out/kt
# 2. the same line without the profiler, and with the full tail
python bench.py --no-extras --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | tail -1 > $out/r03_bench_line_noextras.json
python bench.py --no-extras --no-cpu-baseline --steps 40 --warmup 5 --full-final-conv 2>/dev/null | tail -1 > $out/r03_bench_line_full_final_conv.json
# 3. SQ / HBM counters of the dominant conv (separate passes)
bash tools/pmc_conv.sh $out/pmc_conv > $out/pmc_conv.log 2>&1
cp $out/pmc_conv/conv_pmc.json $out/r03_pmc_conv_raw.json
rm -rf $out/pmc_conv
# 4. HBM counters of K2/K3
bash tools/pmc_warps.sh $out/pmc_warps > /dev/null 2>&1
cp $out/pmc_warps/summary.json $out/r03_pmc_warps.json
rm -rf $out/pmc_warps
# 5. ablations of the dominant conv, same box (standalone back-to-back launches)
for rep in 1 2 3; do
for v in "" TAPS18 TAPS18X2 NOX NOMFMA NOSAT; do
  if [ -z "$v" ]; then lib=$GRAFT_REPO_ROOT/megaportrait-hack_amd/libmphip.so; else lib=$GRAFT_REPO_ROOT/build_variants/libmphip_$v.so; fi
  MPHIP_LIB=$lib python tools/time_one_conv.py 96 96 16 64 64 3 8 1 2>&1 | grep -v amdgpu >> $out/r03_conv_ablations.txt
done
done
# 6. batch sweep through the plan + isolated latencies
python tools/bench_plan.py 2>&1 | grep -v amdgpu > $out/r03_plan_vs_perop.txt
python tools/bench_generator.py 2>&1 | grep -v amdgpu > $out/r03_generator.txt
ls -la $out
