#!/bin/bash
# In-kernel stamps of the hash-backward producer: STAMP_LEVELS="2 8 12 15" bash tools/stamps_produce.sh
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (sets GRAFT_REPO_ROOT)}"
for lv in ${STAMP_LEVELS:-12}; do
cd "$GRAFT_REPO_ROOT/rise_sdf_amd/csrc"
# experiment objects and the variant library live in /tmp/ab and are selected with RSDF_LIB (rise_sdf_amd/_lib.py): the
# shipped rise_sdf_amd/librisesdf_hip.so and _build/ are never overwritten
mkdir -p /tmp/ab; rm -f /tmp/ab/*.o
export RSDF_LIB=/tmp/ab/librisesdf_hip.variant.so
variant_objs() { for o in _build/*.o; do b=$(basename $o); if [ -f /tmp/ab/$b ]; then echo /tmp/ab/$b; else echo $o; fi; done; }
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -I../../include -DRSDF_STAMPS -DRSDF_STAMP_LEVEL=$lv -c hashgrid_fd7.hip -o /tmp/ab/hashgrid_fd7.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(variant_objs) -o $RSDF_LIB
cd ../..
python3 - $lv <<'PY'
import ctypes, sys, io, contextlib, runpy
lv = sys.argv[1]
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--cpu-rays", "0", "--no-extras", "--streams", "1", "--width", "400", "--height", "400", "--no-kernel-timing"]
with contextlib.redirect_stdout(io.StringIO()):
    try:
        runpy.run_path("bench.py", run_name="__main__")
    except SystemExit:
        pass
from rise_sdf_amd import _lib
raw = ctypes.CDLL(_lib.lib()._name)
out = (ctypes.c_ulonglong * 16)()
fn = raw.rsdf_debug_read_pstamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(out, 1) == 0
n = max(int(out[15]), 1)
names = {0: "loads + phase-1 weights", 1: "work-list append", 2: "run merge", 3: "corner indices", 4: "slot atomics (LDS)", 5: "barrier A",
         6: "scan + reservation", 7: "barrier B", 8: "staging writes", 9: "barrier C / D", 10: "copy-out", 11: "phase-2 item evaluation"}
tot = sum(int(out[i]) for i in names)
print("level", lv, "workgroups", n, "cycles per workgroup %.0f" % (tot / n))
for i in sorted(names):
    print("  %-28s %8.0f  %5.1f %%" % (names[i], out[i] / n, 100.0 * out[i] / tot))
PY
done
