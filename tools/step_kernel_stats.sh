#!/bin/bash
# rocprofv3 --kernel-trace --stats of the training step at its operating point (tools/bench_step.py): every kernel with its
# calls and average duration, per step.  bash tools/step_kernel_stats.sh [out.txt]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/sks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sks -o t -- python3 tools/bench_step.py --steps 40 --settle 80 > /tmp/sks.out 2>&1
f=$(find /tmp/sks -name "*kernel_stats.csv")
python3 - "$f" <<'PY' > ${1:-/dev/stdout}
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 80.0 + 40 + 40 + 1   # settle + measured + instrumented (+ the sync-count step): per-step figures are approximate
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per step ~ %.2f ms over %d kernels" % (tot / 1e6 / steps, len(rows)))
for r in rows[:70]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:150]
    print("%-152s calls/step %6.1f  avg %8.1f us  ms/step %6.3f  min %8.1f max %9.1f us" % (
        name, float(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6 / steps,
        float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
