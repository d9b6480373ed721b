cd $GRAFT_REPO_ROOT
export PYTHONFAULTHANDLER=1
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_dropin.py tests/test_gpu_model.py tests/test_gpu_texture.py tests/test_gpu_layer_bwd.py tests/test_gpu_edges.py tests/test_gpu_bf16.py -q -x 2>&1 | tail -3
python bench.py --steps 1 --warmup 1 --cpu-rays 0 --only-extras dropin_path 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['secondary']
for k,v in s.items(): print(k, {kk:vv for kk,vv in v.items() if kk in ('samples_per_s','ms_per_step','top_ms_per_step','error')})"
