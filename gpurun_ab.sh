cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_plan.py tests/test_gpu_backward.py -q -x 2>&1 | grep -E "passed|failed|Error|error" | head -5
for i in 1 2; do
python tools/bench_plan.py 2>&1 | grep "^B=8"
MPHIP_GN_EPILOGUE=0 python tools/bench_plan.py 2>&1 | grep "^B=8"
done
out=gpurun_out/r03_b1; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $out/kt -- python tools/run_plan_steps.py 8 30 > $out/log8.txt 2>&1
db=$(find $out/kt -name "*.db" | head -1)
python tools/lane_timeline.py $db 4.6 > $out/timeline_b8_gnep1.txt
rm -rf $out/kt
grep gn_tile_finalize $out/timeline_b8_gnep1.txt | head -8
