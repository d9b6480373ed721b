cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04cc
export PYTHONFAULTHANDLER=1
python bench.py --steps 1 --warmup 1 --cpu-rays 0 --only-extras dropin_path,c2_800,c3_step 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['secondary']
for k,v in s.items(): print(k, {kk:vv for kk,vv in v.items() if kk in ('samples_per_s','ms_per_step','top_ms_per_step','top','error')})" | tee gpurun_out/r04cc/extras.txt
timeout 1800 python -m pytest tests -q -x -m gpu 2>&1 | tail -6 | tee gpurun_out/r04cc/tests_gpu_tail.log
