#!/bin/bash
# A/B builds of envlight.hip with different -D flags on one box: bash tools/ab_env.sh "-DX=1" "-DX=2" ...
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (sets GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT/rise_sdf_amd/csrc"
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -I../../include $v -c envlight.hip -o _build/envlight.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 _build/*.o -o ../librisesdf_hip.so
  (cd ../.. && bash tools/spec_levels.sh | head -6 | awk '{print $(NF-1)}' | tr '\n' ' '; echo " <= [$v]")
done
