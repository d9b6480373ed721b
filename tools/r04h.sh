cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04h
python tools/bench_c2.py --width 400 --height 400 --steps 2 2>/dev/null | tail -1 > gpurun_out/r04h/c2_400.json
rm -rf /tmp/c2p
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c2p -o t -- python3 tools/bench_c2.py --width 400 --height 400 --steps 2 > /dev/null 2>&1
f=$(find /tmp/c2p -name "*kernel_stats.csv")
python3 - "$f" <<'PY' > gpurun_out/r04h/c2_400_kernel_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 3.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per step ~ %.2f ms over %d kernels (3 steps incl. warm-up)" % (tot / 1e6 / steps, len(rows)))
for r in rows[:70]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:90]
    print("%-92s calls/step %7.1f  avg %8.1f us  ms/step %8.3f" % (name, float(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3,
                                                              float(r["TotalDurationNs"]) / 1e6 / steps))
PY
cat gpurun_out/r04h/c2_400.json
head -50 gpurun_out/r04h/c2_400_kernel_stats.txt
