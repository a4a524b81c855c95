cd /root/repo
python -m pytest tests/test_gpu_plan.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
python tools/bench_plan.py 2>&1 | grep -v amdgpu
python bench.py --no-extras --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['launches_timed'], d['roofline']['demand_driven_launch']['launch_ms'])"
