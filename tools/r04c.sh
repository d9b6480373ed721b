cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_model_fixtures.py -q -x -s -k l16 2>&1 | grep -E "max abs|passed|failed|^E " | head -20
