#!/bin/bash
# usage: tools/bench_short.sh label [env...]: one short bench line (fps, ms/step, dominant launch ms)
label=$1; shift
env "$@" python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"
