#!/bin/bash
# A/B two builds of mlp_fused.hip on the same box: bash tools/ab_flags.sh "<extra flags A>" "<extra flags B>"
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (sets GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT/rise_sdf_amd/csrc"
# experiment objects and the variant library live in /tmp/ab and are selected with RSDF_LIB (rise_sdf_amd/_lib.py): the
# shipped rise_sdf_amd/librisesdf_hip.so and _build/ are never overwritten
mkdir -p /tmp/ab; rm -f /tmp/ab/*.o
export RSDF_LIB=/tmp/ab/librisesdf_hip.variant.so
variant_objs() { for o in _build/*.o; do b=$(basename $o); if [ -f /tmp/ab/$b ]; then echo /tmp/ab/$b; else echo $o; fi; done; }
for v in "$1" "$2" "$1" "$2"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -I../../include $v -c mlp_fused.hip -o /tmp/ab/mlp_fused.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(variant_objs) -o $RSDF_LIB
  (cd ../.. && python bench.py --steps 2 --warmup 1 --cpu-rays 0 --no-extras --streams 1 --width 400 --height 400 2>&1 | tail -1 | grep -o "rsdf_sdfmlp_fd7_fwd[^}]*}\|rsdf_sdfmlp_fd7_bwd\": {[^}]*}" | tr '\n' ' '; echo " <= [$v]")
done
