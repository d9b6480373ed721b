set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
timeout 900 python -m pytest tests/test_gpu_x3.py -x -q 2>&1 | tail -30 > gpurun_out/r04a/x3_tests.log
cat gpurun_out/r04a/x3_tests.log | tail -15
for f in 0 1; do
  RSDF_X3=$f timeout 600 python bench.py --steps 3 --warmup 1 --cpu-rays 0 --no-extras --streams 1 2>/dev/null | tail -1 > gpurun_out/r04a/bench_x3_$f.json
  python3 - <<PY
import json
d=json.load(open("gpurun_out/r04a/bench_x3_$f.json"))
kb=d['kernel_breakdown']
print("X3=$f", '%.4g'%d['value'], {k:round(v['ms_per_step']/v['calls']*d['steps'],2) for k,v in kb.items() if v['ms_per_step']>5})
PY
done
