cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04m
export PYTHONFAULTHANDLER=1
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_layer_bwd.py tests/test_gpu_texture.py tests/test_gpu_bf16.py tests/test_gpu_dropin.py -q -x 2>&1 | tail -4 | tee gpurun_out/r04m/tests.log
python tools/bench_c2.py --width 400 --height 400 --steps 2 2>/dev/null | tail -1 | tee gpurun_out/r04m/c2_400.json
python bench.py --steps 1 --warmup 1 --cpu-rays 0 --only-extras dropin_path 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value']); print(d['secondary'])" | tee gpurun_out/r04m/dropin.txt
