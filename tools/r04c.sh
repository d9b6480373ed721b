cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_layer_bwd.py tests/test_gpu_texture.py -q -x 2>&1 | tail -3
for m in vec mfma; do RSDF_BWD_WEIGHT=$m timeout 600 python tools/bench_step.py --steps 40 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$m', round(d['ms_per_step'],2), 'ms/step; kernels', d['rsdf_kernel_ms_per_step'], {k:v for k,v in d['top'].items() if 'weight' in k})"; done
