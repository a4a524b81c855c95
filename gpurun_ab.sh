cd /root/repo
python -m pytest tests/test_gpu_plan.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
python tools/bench_plan.py 2>&1 | grep -v amdgpu
