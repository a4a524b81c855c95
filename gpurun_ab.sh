cd $GRAFT_REPO_ROOT
( time python bench.py 2>gpurun_out/bench_err.log ) > gpurun_out/r03_bench_default.log 2>&1
tail -4 gpurun_out/r03_bench_default.log | cut -c1-300
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r03_bench_default.log') if x.startswith('{')][-1]
d=json.loads(l)
print(d['value'], d['ms_per_step'], d['config']['batches_in_flight'], d.get('one_in_flight'), d.get('full_final_conv'), d['roofline']['launch_ms'], d['roofline']['frac'])
print({k:(v.get('value') if isinstance(v,dict) else v) for k,v in d.items() if k in ('train_step','reenact_1x64','end_to_end','end_to_end_autocast_fp16','fp32_exact')})
print(d.get('leg_seconds'))
PY
