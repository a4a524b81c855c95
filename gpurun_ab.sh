cd $GRAFT_REPO_ROOT
for i in 1 2; do
for pr in "" low high; do
env ${pr:+MPHIP_PLAN_SIDE_PRIORITY=$pr} python bench.py --no-extras --no-cpu-baseline --steps 40 --warmup 6 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('side prio [$pr]', d['value'], d['ms_per_step'], 'conv', d['roofline']['launch_ms'])"
done
done
