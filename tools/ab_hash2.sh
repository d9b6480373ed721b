#!/bin/bash
# A/B builds of hashgrid_fd7.hip on one box, timed per kernel with tools/bench_hash_fd7.py; the variants go to a separate
# library (RSDF_LIB) so the shipped librisesdf_hip.so is never overwritten:  bash tools/ab_hash2.sh "-DX=1" "-DX=2" ...
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (sets GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT/rise_sdf_amd/csrc"
mkdir -p /tmp/ab
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -I../../include $v -c ${AB_SRC:-hashgrid_fd7.hip} -o /tmp/ab/hashgrid_fd7.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls _build/*.o | grep -v hashgrid_fd7.o) /tmp/ab/hashgrid_fd7.o -o /tmp/ab/librisesdf_hip.ab.so
  (cd ../.. && RSDF_LIB=/tmp/ab/librisesdf_hip.ab.so python tools/bench_hash_fd7.py ${AB_ARGS:-} 2>/dev/null | tail -1; echo " <= [$v] ${AB_SRC:-}")
done
