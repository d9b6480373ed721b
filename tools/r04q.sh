cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04q
export PYTHONFAULTHANDLER=1
for cfg in "2 8192" "2 16384" "1 16384"; do set -- $cfg
  PYTORCH_HIP_ALLOC_CONF=roundup_power2_divisions:4 python tools/bench_c2.py --streams $1 --chunk $2 --steps 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c2 streams $1 chunk $2', d['samples_per_s'], d['ms_per_step'])"
done
for cfg in "2 32768" "3 20480" "3 28672" "2 28672"; do set -- $cfg
  python bench.py --steps 3 --warmup 1 --cpu-rays 0 --no-extras --streams $1 --chunk $2 --no-kernel-timing 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c1 streams $1 chunk $2', d['value'], d['ms_per_step'], d['config']['hbm_gib'])"
done
