cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_x2.py -q -s -x 2>&1 | tail -6
