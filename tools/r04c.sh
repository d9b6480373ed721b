cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_x2.py -q -s 2>&1 | tail -12
for x in 1 0; do RSDF_X2=$x python bench.py --steps 2 --warmup 1 --cpu-rays 0 --no-extras --streams 1 --width 400 --height 400 --hidden 128 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
kb=d['kernel_breakdown']
print('X2=$x H=128', {k:round(v['ms_per_step']/v['calls']*d['steps'],2) for k,v in kb.items() if v['ms_per_step']>10}, '%.4g'%d['value'])"; done
