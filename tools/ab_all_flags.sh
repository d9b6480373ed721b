#!/bin/bash
# Whole-library A/B of extra compiler flags on one box: every translation unit rebuilt with "$1" into /tmp/ab_all, then
# the prefilter alone, a training step and a 400x400 config[2] step on the shipped library and on the variant.
#   bash tools/ab_all_flags.sh "-fno-slp-vectorize"
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (sets GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT/rise_sdf_amd/csrc"
mkdir -p /tmp/ab_all
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -w $1"
for s in core march composite hashgrid hashgrid_fd7 hashgrid_dx mlp mlp_fused mlp_coop mlp_quad mlp_x2 mlp_pair mlp_layer_bwd neus texture gridsample envlight frontend loss; do
  /opt/rocm/bin/hipcc $F -c $s.hip -o /tmp/ab_all/$s.o &
done
for s in mlp mlp_fused mlp_coop mlp_quad mlp_layer_bwd; do
  /opt/rocm/bin/hipcc $F -DRSDF_BF16 -c $s.hip -o /tmp/ab_all/${s}_bf16.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/ab_all/*.o -o /tmp/ab_all/librisesdf_hip.variant.so
cd ../..
run() {
  python tools/bench_prefilter.py 2>/dev/null | tail -1 | cut -c1-200
  python tools/bench_step.py --steps 30 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c3_step %.2f ms' % d['ms_per_step'], dict(list(d.get('top',{}).items())[:8]))"
}
echo "== shipped"; run
echo "== variant [$1]"; RSDF_LIB=/tmp/ab_all/librisesdf_hip.variant.so run
echo "== shipped"; run
echo "== variant [$1]"; RSDF_LIB=/tmp/ab_all/librisesdf_hip.variant.so run
