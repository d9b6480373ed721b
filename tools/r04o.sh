cd $GRAFT_REPO_ROOT
show() { tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['secondary']
print('value', d['value'], 'ms', d['ms_per_step'], d['config']['hbm_gib'])
for k,v in s.items(): print(' ', k, {kk:vv for kk,vv in v.items() if kk in ('samples_per_s','error','hbm_gib','wall_s')})"; }
for conf in "roundup_power2_divisions:4" "roundup_power2_divisions:8" ; do
  echo "== $conf"
  PYTORCH_HIP_ALLOC_CONF=$conf PYTORCH_CUDA_ALLOC_CONF=$conf python bench.py --steps 3 --warmup 1 --cpu-rays 0 --only-extras one_stream,h128,c2_800 2>gpurun_out/r04o_err.txt | show
  grep -i "alloc_conf\|not supported\|unrecognized" gpurun_out/r04o_err.txt | head -3
done
