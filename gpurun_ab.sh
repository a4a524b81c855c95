cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_plan.py tests/test_gpu_backward.py -q -x 2>&1 | grep -E "passed|failed" > gpurun_out/pt.log
cat gpurun_out/pt.log
python tools/bench_plan.py 2>&1 | grep "^B="
MPHIP_F16X3_OLD_SPLITS=1 python tools/bench_plan.py 2>&1 | grep "^B="
python tools/sweep_conv_plans.py 8 1 2>&1 | grep -v amdgpu > gpurun_out/r03_conv_plan_sweep.txt
cut -c1-400 gpurun_out/r03_conv_plan_sweep.txt
