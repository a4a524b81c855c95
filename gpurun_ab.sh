cd /root/repo
python -m pytest tests/test_gpu_plan.py -x -q -m gpu 2>&1 | tail -5
python tools/bench_plan.py 2>&1 | grep -v amdgpu
MPHIP_FULL_FINAL_CONV=1 python tools/bench_plan.py 2>&1 | grep -v amdgpu
