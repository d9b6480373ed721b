#!/bin/bash
# A/B builds of hashgrid_fd7.hip with different -D flags on one box: bash tools/ab_hash.sh "-DX=1" "-DX=2" ...
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (sets GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT/rise_sdf_amd/csrc"
# experiment objects and the variant library live in /tmp/ab and are selected with RSDF_LIB (rise_sdf_amd/_lib.py): the
# shipped rise_sdf_amd/librisesdf_hip.so and _build/ are never overwritten
mkdir -p /tmp/ab; rm -f /tmp/ab/*.o
export RSDF_LIB=/tmp/ab/librisesdf_hip.variant.so
variant_objs() { for o in _build/*.o; do b=$(basename $o); if [ -f /tmp/ab/$b ]; then echo /tmp/ab/$b; else echo $o; fi; done; }
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -I../../include $v -c hashgrid_fd7.hip -o /tmp/ab/hashgrid_fd7.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(variant_objs) -o $RSDF_LIB
  (cd ../.. && python bench.py --steps 2 --warmup 1 --cpu-rays 0 --no-extras --streams 1 --width ${AB_SIZE:-400} --height ${AB_SIZE:-400} 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); kb=d['kernel_breakdown']
print({k:round(v['ms_per_step']/v['calls']*d['steps'],2) for k,v in kb.items() if 'hashgrid' in k}, '%.4g'%d['value'])"; echo " <= [$v]")
done
