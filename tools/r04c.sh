cd $GRAFT_REPO_ROOT
for cfg in "2 28672" "3 18432" "3 20480" "2 32768" "4 14336"; do set -- $cfg; python bench.py --steps 3 --warmup 1 --cpu-rays 0 --no-extras --no-kernel-timing --streams $1 --chunk $2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('streams $1 chunk $2', '%.4g'%d['value'])"; done
