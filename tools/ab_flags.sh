#!/bin/bash
# A/B two builds of mlp_fused.hip on the same box: bash tools/ab_flags.sh "<extra flags A>" "<extra flags B>"
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (sets GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT/rise_sdf_amd/csrc"
for v in "$1" "$2" "$1" "$2"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -I../../include $v -c mlp_fused.hip -o _build/mlp_fused.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 _build/*.o -o ../librisesdf_hip.so
  (cd ../.. && python bench.py --steps 2 --warmup 1 --cpu-rays 0 --width 400 --height 400 2>&1 | tail -1 | grep -o "rsdf_sdfmlp_fd7_fwd[^}]*}\|rsdf_sdfmlp_fd7_bwd\": {[^}]*}" | tr '\n' ' '; echo " <= [$v]")
done
