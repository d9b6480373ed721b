#!/bin/bash
# A/B builds of one translation unit (AB_SRC, default mlp_x2) with different -D flags on one box, timed with bench.py's
# per-entry-point breakdown (one chunk stream, 400x400 view):  bash tools/ab_x2.sh "<bench args>" "-DX" "-DY -DZ" ...
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (sets GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT/rise_sdf_amd/csrc"
SRC=${AB_SRC:-mlp_x2}
mkdir -p /tmp/ab; rm -f /tmp/ab/*.o
export RSDF_LIB=/tmp/ab/librisesdf_hip.variant.so
variant_objs() { for o in _build/*.o; do b=$(basename $o); if [ -f /tmp/ab/$b ]; then echo /tmp/ab/$b; else echo $o; fi; done; }
ARGS="$1"; shift
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC $v -c $SRC.hip -o /tmp/ab/$SRC.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(variant_objs) -o $RSDF_LIB
  (cd ../.. && python bench.py --steps 2 --warmup 1 --cpu-rays 0 --no-extras --streams 1 --width 400 --height 400 $ARGS 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
kb=d['kernel_breakdown']
print({k:round(v['ms_per_step']/v['calls']*d['steps'],2) for k,v in kb.items() if v['ms_per_step']>20}, '%.4g'%d['value'])"; echo " <= [$SRC $v] $ARGS")
done
