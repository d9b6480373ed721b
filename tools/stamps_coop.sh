#!/bin/bash
# In-kernel stamps of the lean cooperative forward (H = 128): rebuild mlp_coop.hip with -DRSDF_STAMPS on the box, run one
# bench step at --hidden 128 and print the cycles per tile between the stamp points of wave 0 / workgroup 0.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (sets GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT/rise_sdf_amd/csrc"
# experiment objects and the variant library live in /tmp/ab and are selected with RSDF_LIB (rise_sdf_amd/_lib.py): the
# shipped rise_sdf_amd/librisesdf_hip.so and _build/ are never overwritten
mkdir -p /tmp/ab; rm -f /tmp/ab/*.o
export RSDF_LIB=/tmp/ab/librisesdf_hip.variant.so
variant_objs() { for o in _build/*.o; do b=$(basename $o); if [ -f /tmp/ab/$b ]; then echo /tmp/ab/$b; else echo $o; fi; done; }
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -I../../include -DRSDF_STAMPS ${STAMP_FLAGS:-} -c mlp_coop.hip -o /tmp/ab/mlp_coop.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(variant_objs) -o $RSDF_LIB
cd ../..
python3 - <<'PY'
import ctypes, sys, io, contextlib, runpy
sys.argv = ["bench.py", "--hidden", "128", "--steps", "1", "--warmup", "0", "--cpu-rays", "0", "--no-extras", "--streams", "1", "--width", "400", "--height", "400",
            "--no-kernel-timing"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    try:
        runpy.run_path("bench.py", run_name="__main__")
    except SystemExit:
        pass
from rise_sdf_amd import _lib
raw = ctypes.CDLL(_lib.lib()._name) if hasattr(_lib.lib(), "_name") else _lib.lib()
out = (ctypes.c_ulonglong * 16)()
fn = raw.rsdf_debug_read_stamps
fn.argtypes = [ctypes.c_void_p]
assert fn(out) == 0
n = max(int(out[15]), 1)
names = {9: "tail of previous tile", 0: "wait for the DMA (vmcnt)", 1: "X image store + next DMA issue", 2: "barrier (a)",
         3: "layer-1 reads + 18 MFMAs issued", 4: "Softplus + split + H1 store", 5: "H1 barrier", 6: "layer-2 reads + 48 MFMAs issued",
         7: "Softplus (waits for the MFMAs)", 8: "SDF partial + h2c store", 10: "barrier (b)"}
tot = sum(int(out[i]) for i in names)
print("tiles", n, "cycles per tile %.0f" % (tot / n))
for i in (9, 0, 1, 2, 3, 4, 5, 6, 7, 8, 10):
    print("%-36s %8.0f  %5.1f %%" % (names[i], out[i] / n, 100.0 * out[i] / tot))
PY
