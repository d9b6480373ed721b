cd $GRAFT_REPO_ROOT
python bench.py --steps 3 --warmup 1 --cpu-rays 0 --only-extras chunk4096,one_stream,h128,h128_fp16,c2_800 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['secondary']
print('value', d['value'], 'ms', d['ms_per_step'], d['config']['hbm_gib'])
for k,v in s.items(): print(' ', k, {kk:vv for kk,vv in v.items() if kk in ('samples_per_s','error','hbm_gib','wall_s')})"
