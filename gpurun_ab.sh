cd /root/repo
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "upsample or resampl or flowfield or generator or hot_slice_full_golden or matmul" 2>&1 | tail -3
python tools/bench_generator.py 2>&1 | grep -v amdgpu
for tw in 2048 4096; do for mc in 8 4; do echo "target $tw minch $mc"; MPHIP_GATHER_TARGET_WAVES=$tw MPHIP_GATHER_MIN_CH=$mc python tools/bench_generator.py 2>&1 | grep -v amdgpu; done; done
echo "target 4096 minch 4 maxsplits 64"; MPHIP_GATHER_TARGET_WAVES=4096 MPHIP_GATHER_MIN_CH=4 MPHIP_GATHER_MAX_SPLITS=64 python tools/bench_generator.py 2>&1 | grep -v amdgpu
for v in 1 0; do echo "upsample lds=$v"; MPHIP_UPSAMPLE_LDS=$v python bench.py --no-extras --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"; done
