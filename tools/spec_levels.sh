#!/bin/bash
# per-level durations of the specular prefilter kernels inside one c4-shaped step (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/sl
rocprofv3 --kernel-trace --output-format csv -d /tmp/sl -o t -- python3 tools/bench_c4_step.py > /dev/null 2>&1
f=$(find /tmp/sl -name "*kernel_trace.csv")
python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "specular" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-12:]
for r in last:
    name = "bwd" if "<true" in r["Kernel_Name"].replace(" ", "") or "true," in r["Kernel_Name"] else "fwd"
    print(r["Kernel_Name"][:60].replace("(anonymous namespace)::", ""), r.get("Grid_Size_X", r.get("Grid_Size", "?")), "%.3f ms" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
