#!/bin/bash
# A/B builds of one translation unit (AB_SRC, default mlp_pair) with different -D / compiler flags on one box, timed on a
# config[2] step (tools/bench_c2.py, 400x400 view):  bash tools/ab_c2.sh "<bench_c2 args>" "-DX" "-DY -DZ" ...
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (sets GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT/rise_sdf_amd/csrc"
SRC=${AB_SRC:-mlp_pair}
mkdir -p /tmp/ab; rm -f /tmp/ab/*.o
export RSDF_LIB=/tmp/ab/librisesdf_hip.variant.so
variant_objs() { for o in _build/*.o; do b=$(basename $o); if [ -f /tmp/ab/$b ]; then echo /tmp/ab/$b; else echo $o; fi; done; }
ARGS="$1"; shift
# the Makefile's per-unit flags (NOSLP list); AB_BASE="" to build a variant without them
case $SRC in mlp_x2|mlp_pair|hashgrid_fd7) BASE=${AB_BASE--fno-slp-vectorize};; *) BASE=${AB_BASE-};; esac
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC $BASE $v -c $SRC.hip -o /tmp/ab/$SRC.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(variant_objs) -o $RSDF_LIB
  (cd ../.. && python tools/bench_c2.py --width 400 --height 400 --chunk 16384 --steps 2 $ARGS 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(dict(list(d['top'].items())[:9]), '%.4g samples/s' % d['samples_per_s'], '%.1f ms' % d['ms_per_step'])"; echo " <= [$SRC $v] $ARGS")
done
