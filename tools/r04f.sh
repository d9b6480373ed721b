cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04f
export PYTHONFAULTHANDLER=1
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_regimes.py -q -x -k "hashgrid or hash_backward or curvature" 2>&1 | tail -15 | tee gpurun_out/r04f/tests.log
AB_ARGS="--only bwd --forms pts --rays 28672" timeout 600 bash tools/ab_hash2.sh "-DRSDF_REC_FP32" "-DRSDF_NOP=1" 2>&1 | tee gpurun_out/r04f/ab.log
