cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04aa
rm -rf /tmp/dp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dp -o t -- python3 bench.py --steps 1 --warmup 0 --cpu-rays 0 --width 200 --height 200 --only-extras dropin_path > /dev/null 2>&1
f=$(find /tmp/dp -name "*kernel_stats.csv")
python3 - "$f" <<'PY' | tee gpurun_out/r04aa/dropin_kernel_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms %.1f over %d kernels" % (tot / 1e6, len(rows)))
for r in rows[:40]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:100]
    print("%-102s calls %6s  avg %9.1f us  total %8.2f ms  %5.1f %%" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100*float(r["TotalDurationNs"])/tot))
PY
