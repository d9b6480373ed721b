cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04j
export PYTHONFAULTHANDLER=1
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_regimes.py tests/test_gpu_analytic.py -q -x -k "hashgrid or hash_backward or curvature or binned" 2>&1 | tail -5 | tee gpurun_out/r04j/tests.log
AB_ARGS="--only bwd --forms pts --rays 28672" timeout 900 bash tools/ab_hash_prof.sh "-DRSDF_R_UNROLL=4" "-DRSDF_R_UNROLL=8" "-DRSDF_R_UNROLL=2" "-DRSDF_REC_FP32" 2>&1 | tee gpurun_out/r04j/ab.log
