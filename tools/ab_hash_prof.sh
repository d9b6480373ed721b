#!/bin/bash
# per-kernel durations (rocprofv3 kernel trace) of tools/bench_hash_fd7.py for A/B builds of hashgrid_fd7.hip:
#   AB_SRC=<source file in csrc> bash tools/ab_hash_prof.sh "<flags>" ...
set -uo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT/rise_sdf_amd/csrc"
mkdir -p /tmp/ab
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -I../../include $v -c ${AB_SRC:-hashgrid_fd7.hip} -o /tmp/ab/hashgrid_fd7.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls _build/*.o | grep -v hashgrid_fd7.o) /tmp/ab/hashgrid_fd7.o -o /tmp/ab/librisesdf_hip.ab.so
  rm -rf /tmp/abp
  (cd ../.. && RSDF_LIB=/tmp/ab/librisesdf_hip.ab.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abp -o t -- python3 tools/bench_hash_fd7.py ${AB_ARGS:-} > /dev/null 2>&1)
  echo "== [$v] ${AB_SRC:-}"
  python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/abp/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'fd7' in r['Name']:
        print('  %-40s calls %4s  avg %.3f ms' % (r['Name'][:40], r['Calls'], float(r['AverageNs']) / 1e6))
PY
done
