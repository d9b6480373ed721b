cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04v
export PYTHONFAULTHANDLER=1
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_regimes.py tests/test_gpu_x2.py tests/test_gpu_model.py -q -x 2>&1 | tail -3 | tee gpurun_out/r04v/tests.log
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r04v/bench_driver_cmd.json 2> gpurun_out/r04v/bench.err
tail -1 gpurun_out/r04v/bench_driver_cmd.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['secondary']
print(d['value'], d['ms_per_step'], d.get('wall_s'), d['config']['hbm_gib'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])
print({k:round(v['ms_per_step'],1) for k,v in list(d['kernel_breakdown'].items())[:5]})
for k in s:
    v=s[k]; print(k, round(v.get('samples_per_s',0)/1e6,2), round(v.get('ms_per_step',0),2), v.get('error'))"
