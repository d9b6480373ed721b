cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04d
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r04d/bench_driver_cmd.json 2> gpurun_out/r04d/bench_driver_cmd.err
tail -c 1500 gpurun_out/r04d/bench_driver_cmd.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r04d/bench_driver_cmd.json").read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'])
print({k:(v if not isinstance(v,dict) else v.get('value')) for k,v in d.get('secondary',{}).items()})
PY
