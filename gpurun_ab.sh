cd /root/repo
export TMPDIR=/tmp
python -m pytest tests/test_gpu_plan.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
out=gpurun_out/r03_profiles; mkdir -p $out
rocprofv3 --kernel-trace -d $out/kt -- python tools/run_plan_steps.py 8 12 > /dev/null 2>&1
db=$(find $out/kt -name "*.db" | head -1)
python tools/lane_timeline.py $db 4.4 > $out/r03_timeline_plan.txt
python tools/agg_summary.py $db 12 45 > $out/r03_kernel_agg.txt
python tools/rocpd_summary.py $db $out/r03_kernel_stats.csv
rm -rf $out/kt
