cd $GRAFT_REPO_ROOT
for f in 1 2; do
python bench.py --no-cpu-baseline --extras-budget 1 --steps 40 --warmup 6 --inflight $f 2>gpurun_out/bench_err.log | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('inflight', d['config']['batches_in_flight'], d['value'], d['ms_per_step'], 'conv', d['roofline']['launch_ms'], d['roofline']['frac'], d['roofline']['launches_timed'], 'demand', (d['roofline'].get('demand_driven_launch') or {}).get('launch_ms'), 'one_in_flight', (d.get('one_in_flight') or {}).get('dominant_conv'))"
done
python -m pytest tests/test_gpu_plan.py -q -x 2>&1 | tail -2
