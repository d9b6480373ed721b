cd $GRAFT_REPO_ROOT
for m in rows mfma; do RSDF_BWD_WEIGHT=$m python bench.py --steps 1 --warmup 1 --cpu-rays 0 --width 400 --height 400 --streams 1 --only-extras dropin_path 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
c=d['secondary']['dropin_path']
print('$m', '%.4g'%c['samples_per_s'], c['top_ms_per_step'])"; done
