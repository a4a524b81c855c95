cd /root/repo
python -m pytest tests/test_gpu_backward.py tests/test_gpu_plan.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -3
python bench.py --mode train --steps 15 --warmup 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train demand', d['value'], d['ms_per_step'])"
MPHIP_FULL_FINAL_CONV=1 python bench.py --mode train --steps 15 --warmup 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train full', d['value'], d['ms_per_step'])"
python bench.py --mode train --graph 1 --steps 15 --warmup 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train graph demand', d['value'], d['ms_per_step'])"
