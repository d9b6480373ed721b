set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
timeout 900 python -m pytest tests/test_gpu_x2.py -q -s 2>&1 | tail -40 > gpurun_out/r04a/x2_tests.log
cat gpurun_out/r04a/x2_tests.log | tail -25
