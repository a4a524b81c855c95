cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_b1; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $out/kt -- python tools/run_plan_steps.py 8 30 > $out/log8.txt 2>&1
db=$(find $out/kt -name "*.db" | head -1)
python tools/lane_timeline.py $db 4.6 > $out/timeline_b8_k1.txt
rm -rf $out/kt
grep -E "k1_f16x3|gather_kernel<1" $out/timeline_b8_k1.txt | head -12
