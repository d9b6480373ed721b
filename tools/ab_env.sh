#!/bin/bash
# A/B builds of envlight.hip with different -D flags on one box: bash tools/ab_env.sh "-DX=1" "-DX=2" ...
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (sets GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT/rise_sdf_amd/csrc"
# experiment objects and the variant library live in /tmp/ab and are selected with RSDF_LIB (rise_sdf_amd/_lib.py): the
# shipped rise_sdf_amd/librisesdf_hip.so and _build/ are never overwritten
mkdir -p /tmp/ab; rm -f /tmp/ab/*.o
export RSDF_LIB=/tmp/ab/librisesdf_hip.variant.so
variant_objs() { for o in _build/*.o; do b=$(basename $o); if [ -f /tmp/ab/$b ]; then echo /tmp/ab/$b; else echo $o; fi; done; }
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -I../../include $v -c envlight.hip -o /tmp/ab/envlight.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(variant_objs) -o $RSDF_LIB
  (cd ../.. && bash tools/spec_levels.sh | head -6 | awk '{print $(NF-1)}' | tr '\n' ' '; echo " <= [$v]")
done
