#!/bin/bash
# GPU-busy time, launch count and idle gaps of the c4-shaped training step (rocprofv3 kernel trace of tools/bench_c4_step.py)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python3 tools/bench_c4_step.py --steps 20 > /tmp/tl.out 2>&1
tail -1 /tmp/tl.out | cut -c1-120
f=$(find /tmp/tl -name "*kernel_trace.csv")
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 20 steps: find step boundaries by the Adam kernels? simpler: take the last 60 % of the trace and divide by steps
n = len(rows)
sel = rows[int(n * 0.45):]
t0, t1 = int(sel[0]["Start_Timestamp"]), int(sel[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel)
print("kernels in window: %d, window %.1f ms, GPU busy %.1f ms (%.0f %%)" % (len(sel), (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0)))
c = collections.Counter()
d = collections.Counter()
for r in sel:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:70]
    c[k] += 1
    d[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, v in d.most_common(25):
    print("%-72s %6d calls %8.2f ms" % (k, c[k], v / 1e6))
PY
