#!/bin/bash
# A/B builds of hashgrid.hip for the generic gather: bash tools/ab_gather.sh "-DX=1" "-DX=2" ...   (each variant runs
# tools/bench_gather.py; RSDF_STAGED_BATCH in the environment is passed through)
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (sets GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT/rise_sdf_amd/csrc"
mkdir -p /tmp/ab; rm -f /tmp/ab/*.o
export RSDF_LIB=/tmp/ab/librisesdf_hip.variant.so
variant_objs() { for o in _build/*.o; do b=$(basename $o); if [ -f /tmp/ab/$b ]; then echo /tmp/ab/$b; else echo $o; fi; done; }
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -I../../include $v -c hashgrid.hip -o /tmp/ab/hashgrid.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(variant_objs) -o $RSDF_LIB
  echo "== [$v] batch=${RSDF_STAGED_BATCH:-default}"
  (cd ../.. && python tools/bench_gather.py 2>/dev/null | grep "n= *\(4194304\|15261002\)")
done
