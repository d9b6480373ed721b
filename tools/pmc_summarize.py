"""Reduce a rocprofv3 --pmc csv directory to per-kernel sums: kernel, dispatches, <counter>...  (tools/pmc_passes.sh)."""
import csv
import glob
import os
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
files = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
if not files:
    print("no counter_collection.csv under", src, [f for f in glob.glob(os.path.join(src, "**", "*"), recursive=True)][:20])
    sys.exit(0)
acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for f in files:
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "?")
        k = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-90:]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        disp[k].add(row.get("Dispatch_Id"))
counters = sorted({c for v in acc.values() for c in v})
with open(dst, "w") as o:
    w = csv.writer(o)
    w.writerow(["kernel", "dispatches"] + counters)
    for k in sorted(acc, key=lambda k: -max(acc[k].values())):
        w.writerow([k, len(disp[k])] + [acc[k].get(c, 0.0) for c in counters])
print("wrote", dst, len(acc), "kernels")
