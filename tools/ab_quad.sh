#!/bin/bash
# A/B builds of mlp_quad.hip with different -D flags on one box: bash tools/ab_coop.sh "<bench args>" "-DX" "-DY -DZ" ...
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (sets GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT/rise_sdf_amd/csrc"
ARGS="$1"; shift
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC $v -c mlp_quad.hip -o _build/mlp_quad.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 _build/*.o -o ../librisesdf_hip.so
  (cd ../.. && python bench.py --steps 2 --warmup 1 --cpu-rays 0 --width 400 --height 400 $ARGS 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
kb=d['kernel_breakdown']
print({k:round(v['ms_per_step']/v['calls']*d['steps'],2) for k,v in kb.items() if 'sdfmlp' in k}, '%.4g'%d['value'])"; echo " <= [$v] $ARGS")
done
