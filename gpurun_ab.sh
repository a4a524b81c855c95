cd /root/repo
python -m pytest tests/test_gpu_parity.py tests/test_gpu_plan.py -x -q -m gpu -k "warp or plain_c or full_size or golden or c_abi" 2>&1 | grep -E "passed|failed|Error" | tail -3
for v in 1 0 1 0; do echo "VEC4=$v"; MPHIP_K2_VEC4=$v python tools/bench_warps.py 8 20 2>&1 | grep -v amdgpu | grep "K2"; done
