#!/bin/bash
# A/B of one translation unit (AB_SRC, default mlp) on the drop-in path (bench.py secondary.dropin_path: the reference-shaped
# per-operator route, 4096-ray chunks):  bash tools/ab_dropin.sh "-DX" "-fno-slp-vectorize" ...
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (sets GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT/rise_sdf_amd/csrc"
SRC=${AB_SRC:-mlp}
mkdir -p /tmp/ab; rm -f /tmp/ab/*.o
export RSDF_LIB=/tmp/ab/librisesdf_hip.variant.so
variant_objs() { for o in _build/*.o; do b=$(basename $o); if [ -f /tmp/ab/$b ]; then echo /tmp/ab/$b; else echo $o; fi; done; }
case $SRC in mlp_x2|mlp_pair|hashgrid_fd7) BASE=${AB_BASE--fno-slp-vectorize};; *) BASE=${AB_BASE-};; esac
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC $BASE $v -c $SRC.hip -o /tmp/ab/$SRC.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(variant_objs) -o $RSDF_LIB
  (cd ../.. && python bench.py --steps 2 --warmup 1 --cpu-rays 0 --only-extras dropin_path 2>/dev/null | tail -1 | python3 -c "
import json,sys
s=json.loads(sys.stdin.read())['secondary']['dropin_path']
print(s['top_ms_per_step'], '%.4g' % s['samples_per_s'])"; echo " <= [$SRC $v]")
done
