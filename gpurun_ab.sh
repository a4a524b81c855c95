cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_plan.py -q -x 2>&1 | tail -5
