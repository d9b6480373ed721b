cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c
timeout 1500 python -m pytest tests/test_gpu_convergence.py -q -s -x 2>&1 | grep -E "held-out|SGD|Adam|passed|failed" | tee gpurun_out/r04c/convergence.log
