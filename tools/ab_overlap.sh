#!/bin/bash
# tools/overlap_probe.py against builds of mlp_fused.hip / hashgrid_fd7.hip with different flags (separate library, RSDF_LIB):
#   bash tools/ab_overlap.sh "<mlp flags>|<hash flags>" ...
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (sets GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT/rise_sdf_amd/csrc"
mkdir -p /tmp/ab
CC="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -I../../include"
for v in "$@"; do
  mf="${v%%|*}"; hf="${v##*|}"
  $CC $mf -c mlp_fused.hip -o /tmp/ab/mlp_fused.o
  $CC $hf -c hashgrid_fd7.hip -o /tmp/ab/hashgrid_fd7.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls _build/*.o | grep -v -e hashgrid_fd7.o -e mlp_fused.o) /tmp/ab/mlp_fused.o /tmp/ab/hashgrid_fd7.o -o /tmp/ab/librisesdf_hip.ab.so
  (cd ../.. && RSDF_LIB=/tmp/ab/librisesdf_hip.ab.so python tools/overlap_probe.py ${AB_ARGS:-} 2>/dev/null | tail -1; echo " <= [$v]")
done
