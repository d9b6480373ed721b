cd $GRAFT_REPO_ROOT
timeout 600 python tools/bench_step.py --aten-sites 2>/dev/null | tail -45
