cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04f
( time python bench.py --steps 20 --warmup 5 > gpurun_out/r04f/bench.json 2> gpurun_out/r04f/bench.err ) 2> gpurun_out/r04f/time.txt; cat gpurun_out/r04f/time.txt

python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r04f/bench.json").read().strip().splitlines()[-1])
print('%.4g'%d['value'], round(d['ms_per_step'],1))
for k,v in d['secondary'].items():
    if isinstance(v,dict): print(k, '%.4g'%v.get('samples_per_s',0), round(v.get('ms_per_step',0),2), v.get('error',''))
PY
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r04f/bench.json").read().strip().splitlines()[-1])
print(d.get('wall_s')); print({k:v.get('wall_s') for k,v in d['secondary'].items() if isinstance(v,dict)})
PY
