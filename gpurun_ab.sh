cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -x -s -k "framework_initialisation" 2>&1 | grep -E "default init|passed|failed|Error|assert" | head
