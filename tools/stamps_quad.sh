#!/bin/bash
# In-kernel stamps of the quad backward (H = 64): rebuild mlp_quad.hip with -DRSDF_STAMPS on the box, run one bench step
# and print the cycles per tile between the stamp points of wave 0 / workgroup 0.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (sets GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT/rise_sdf_amd/csrc"
# experiment objects and the variant library live in /tmp/ab and are selected with RSDF_LIB (rise_sdf_amd/_lib.py): the
# shipped rise_sdf_amd/librisesdf_hip.so and _build/ are never overwritten
mkdir -p /tmp/ab; rm -f /tmp/ab/*.o
export RSDF_LIB=/tmp/ab/librisesdf_hip.variant.so
variant_objs() { for o in _build/*.o; do b=$(basename $o); if [ -f /tmp/ab/$b ]; then echo /tmp/ab/$b; else echo $o; fi; done; }
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -I../../include -DRSDF_STAMPS ${STAMP_FLAGS:-} -c mlp_quad.hip -o /tmp/ab/mlp_quad.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(variant_objs) -o $RSDF_LIB
cd ../..
python3 - <<'PY'
import ctypes, sys, io, contextlib, runpy
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--cpu-rays", "0", "--no-extras", "--streams", "1", "--width", "400", "--height", "400", "--no-kernel-timing"]
with contextlib.redirect_stdout(io.StringIO()):
    try:
        runpy.run_path("bench.py", run_name="__main__")
    except SystemExit:
        pass
from rise_sdf_amd import _lib
raw = ctypes.CDLL(_lib.lib()._name)
out = (ctypes.c_ulonglong * 16)()
fn = raw.rsdf_debug_read_qstamps
fn.argtypes = [ctypes.c_void_p]
assert fn(out) == 0
n = max(int(out[15]), 1)
names = {0: "loop overhead", 1: "vmcnt wait (inputs landed, stores retired)", 2: "X staging + d_sdf loads + next DMA issue", 3: "barrier 1",
         4: "layer-1 recompute, Softplus, H1 store", 5: "barrier 2", 6: "layer-2 recompute, dz2, store", 7: "barrier 3",
         8: "dz1 and dW1 products, dz1 store", 9: "barrier 4", 10: "dx + plane stores, dW0 products"}
tot = sum(int(out[i]) for i in names)
print("tiles", n, "cycles per tile %.0f" % (tot / n))
for i in sorted(names):
    print("%-46s %8.0f  %5.1f %%" % (names[i], out[i] / n, 100.0 * out[i] / tot))
PY
