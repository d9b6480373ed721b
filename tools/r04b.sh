cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04b
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r04b/suite.log
tail -5 gpurun_out/r04b/suite.log
