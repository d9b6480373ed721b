cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04b
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > gpurun_out/r04b/suite.log
tail -15 gpurun_out/r04b/suite.log
