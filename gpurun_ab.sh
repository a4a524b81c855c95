cd $GRAFT_REPO_ROOT
for i in 1 2; do
for q in 4 8 16; do
GPU_MAX_HW_QUEUES=$q python bench.py --no-extras --no-cpu-baseline --steps 40 --warmup 6 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('queues $q inflight', d['config']['batches_in_flight'], d['value'], d['ms_per_step'])"
done
done
GPU_MAX_HW_QUEUES=8 python bench.py --no-extras --no-cpu-baseline --steps 40 --warmup 6 --inflight 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('queues 8 inflight', d['config']['batches_in_flight'], d['value'], d['ms_per_step'])"
GPU_MAX_HW_QUEUES=8 python bench.py --no-extras --no-cpu-baseline --steps 40 --warmup 6 --inflight 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('queues 8 inflight', d['config']['batches_in_flight'], d['value'], d['ms_per_step'])"
