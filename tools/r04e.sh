cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04e
export PYTHONFAULTHANDLER=1
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "opacity or accumulate or large_hashmap" 2>&1 | tail -5
# the order in which the withdrawn fold faulted in round 3 (gpurun_out/r03g/repro.log): test_gpu_ops.py, then the stage-1 model test
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_split_model.py -q -x 2>&1 | tail -30 > gpurun_out/r04e/order_fold.log
tail -8 gpurun_out/r04e/order_fold.log
