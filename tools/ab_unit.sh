#!/bin/bash
# A/B builds of ONE translation unit of rise_sdf_amd/csrc with extra -D flags on one box, each timed with a command of the
# caller's choice; the variants go to a separate library (RSDF_LIB), the shipped one is never overwritten:
#   AB_UNIT=mlp AB_CMD="python tools/bench_linear.py" bash tools/ab_unit.sh "-DX=1" "-DX=2" ...
set -uo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (sets GRAFT_REPO_ROOT)}"
U=${AB_UNIT:?translation unit without .hip}
cd "$GRAFT_REPO_ROOT/rise_sdf_amd/csrc"
mkdir -p /tmp/ab
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -I../../include $v -c $U.hip -o /tmp/ab/$U.o || continue
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls _build/*.o | grep -v "/$U.o") /tmp/ab/$U.o -o /tmp/ab/librisesdf_hip.ab.so
  echo "== [$v] $U"
  (cd ../.. && RSDF_LIB=/tmp/ab/librisesdf_hip.ab.so ${AB_CMD} 2>/dev/null | tail -${AB_TAIL:-1})
done
