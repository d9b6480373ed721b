#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of tools/bench_hash_fd7.py for A/B builds of hashgrid_fd7.hip (separate passes):
#   [AB_SRC=<file in csrc>] bash tools/ab_hash_pmc.sh "<flags>" ...
set -uo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT/rise_sdf_amd/csrc"
mkdir -p /tmp/ab
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -I../../include $v -c ${AB_SRC:-hashgrid_fd7.hip} -o /tmp/ab/hashgrid_fd7.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls _build/*.o | grep -v hashgrid_fd7.o) /tmp/ab/hashgrid_fd7.o -o /tmp/ab/librisesdf_hip.ab.so
  echo "== [$v] ${AB_SRC:-}"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/abp
    (cd ../.. && RSDF_LIB=/tmp/ab/librisesdf_hip.ab.so rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/abp -o t -- python3 tools/bench_hash_fd7.py --forms pts --only bwd --reps 2 > /tmp/abp.log 2>&1)
    python3 - $c <<'PY'
import csv, glob, sys, collections, json, re
c = sys.argv[1]
f = glob.glob('/tmp/abp/**/*counter_collection.csv', recursive=True)[0]
tot = collections.Counter(); n = collections.Counter()
for r in csv.DictReader(open(f)):
    if 'fd7' in r['Kernel_Name'] and r['Counter_Name'] == c:
        k = 'produce' if 'produce' in r['Kernel_Name'] else 'reduce'
        tot[k] += float(r['Counter_Value']); n[k] += 1
S = 21051603.0
try:
    S = float(json.loads([l for l in open('/tmp/abp.log') if l.startswith('{')][-1])['samples'])
except Exception:
    pass
for k in tot:
    b = tot[k] / n[k] * 1024.0 * (2.0 if c == 'FETCH_SIZE' else 1.0)
    print('  %-10s %-8s %8.0f B/sample (%d launches)' % (c, k, b / S, n[k]))
PY
  done
done
