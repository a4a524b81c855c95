cd /root/repo
export TMPDIR=/tmp
for rep in 1 2; do for o in 1 0; do echo "order=$o"; MPHIP_PLAN_CHAIN_ORDER=$o python bench.py --no-extras --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['step_ms'])"; done; done
out=gpurun_out/r03_profiles; mkdir -p $out
rocprofv3 --kernel-trace -d $out/kt -- python tools/run_plan_steps.py 8 12 > /dev/null 2>&1
db=$(find $out/kt -name "*.db" | head -1)
python tools/lane_timeline.py $db 4.4 > $out/r03_timeline_plan_s2cfirst.txt
rm -rf $out/kt
