cd /root/repo
( time python bench.py 2>/dev/null ) > gpurun_out/r03_bench_default.log 2>&1
tail -5 gpurun_out/r03_bench_default.log | cut -c1-400
python -m pytest tests -x -q -m gpu 2>&1 | tail -5
