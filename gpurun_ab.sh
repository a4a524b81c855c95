cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -x -k "groupnorm_statistics or k1_f16x3" 2>&1 | tail -5
