cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04b
RSDF_CHECK=1 timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > gpurun_out/r04b/suite_check.log
tail -6 gpurun_out/r04b/suite_check.log
