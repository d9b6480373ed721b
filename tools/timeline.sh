#!/bin/bash
# GPU-busy time, launch count and idle gaps of any command (rocprofv3 kernel trace):
#   bash tools/timeline.sh <tag> <window_fraction> -- python3 bench.py --chunk 4096 --steps 1 --warmup 1 --cpu-rays 0
# Only the last <window_fraction> of the launches are analysed (skips warm-up); biggest idle gaps are listed with the
# kernels on either side.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=$1; FRAC=$2; shift 3
rm -rf /tmp/tl_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$TAG -o t -- "$@" > /tmp/tl_$TAG.out 2>&1
tail -1 /tmp/tl_$TAG.out | cut -c1-160
f=$(find /tmp/tl_$TAG -name "*kernel_trace.csv")
python3 - "$f" "$FRAC" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
sel = rows[int(n * (1.0 - float(sys.argv[2]))):]
t0, t1 = int(sel[0]["Start_Timestamp"]), int(sel[-1]["End_Timestamp"])
name = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:48]
busy = 0
end = t0
gaps = []
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > end:
        gaps.append((s - end, prev, name(r)))
    busy += max(0, e - max(s, end))
    if e > end:
        end, prev = e, name(r)
print("kernels in window: %d, window %.1f ms, GPU busy %.1f ms (%.1f %%), idle %.1f ms in %d gaps" %
      (len(sel), (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), sum(g[0] for g in gaps) / 1e6, len(gaps)))
by = collections.Counter()
cnt = collections.Counter()
for g, a, b in gaps:
    by[(a, b)] += g
    cnt[(a, b)] += 1
for (a, b), v in by.most_common(14):
    print("  idle %8.2f ms in %5d gaps: after %-48s before %s" % (v / 1e6, cnt[(a, b)], a, b))
c = collections.Counter()
d = collections.Counter()
for r in sel:
    k = name(r)
    c[k] += 1
    d[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, v in d.most_common(16):
    print("%-50s %6d calls %9.2f ms" % (k, c[k], v / 1e6))
PY
