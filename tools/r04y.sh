cd $GRAFT_REPO_ROOT
export PYTHONFAULTHANDLER=1
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_layer_bwd.py tests/test_gpu_texture.py tests/test_gpu_edges.py tests/test_gpu_bf16.py -q -x 2>&1 | tail -3
echo "== TS on"; python tools/bench_linear.py --rows 8000000 --shapes 128x128,84x128 2>/dev/null | tail -1
echo "== TS off"; RSDF_LIN_TS=0 python tools/bench_linear.py --rows 8000000 --shapes 128x128,84x128 2>/dev/null | tail -1
echo "== TS on, 250k rows"; python tools/bench_linear.py --rows 250000 --reps 50 --shapes 128x128 2>/dev/null | tail -1
echo "== TS off, 250k rows"; RSDF_LIN_TS=0 python tools/bench_linear.py --rows 250000 --reps 50 --shapes 128x128 2>/dev/null | tail -1
