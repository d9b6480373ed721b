cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04i
export PYTHONFAULTHANDLER=1
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_regimes.py tests/test_gpu_analytic.py -q -x -k "hashgrid or hash_backward or curvature or binned" 2>&1 | tail -8 | tee gpurun_out/r04i/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-extras > gpurun_out/r04i/bench_noextras.json 2> gpurun_out/r04i/bench.err
tail -1 gpurun_out/r04i/bench_noextras.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])
print({k:round(v['ms_per_step'],1) for k,v in list(d['kernel_breakdown'].items())[:6]})"
