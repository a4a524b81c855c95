cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_ss; rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $out/kt -- python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 --single-stream-plan > $out/log.txt 2>&1
db=$(find $out/kt -name "*.db" | head -1)
python tools/lane_timeline.py $db 8.0 > $out/timeline_ss2.txt
rm -rf $out/kt
