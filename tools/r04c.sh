cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sync_free.py -q -x 2>&1 | tail -3
