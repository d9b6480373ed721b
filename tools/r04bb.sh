cd $GRAFT_REPO_ROOT
export PYTHONFAULTHANDLER=1
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_dropin.py tests/test_gpu_regimes.py -q -x 2>&1 | tail -3
python bench.py --steps 1 --warmup 1 --cpu-rays 0 --only-extras dropin_path 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value']); print(d['secondary'])"
