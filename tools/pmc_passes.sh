#!/bin/bash
# PMC counter passes for bench.py (one counter group per run, kernel-trace only; MI355X_MICROARCH.md "rocprofv3
# PMC slots").  Run on the GPU box: bash tools/pmc_passes.sh [bench args].  Summaries land in gpurun_out/pmc/.
set -uo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
ARGS=${@:---steps 1 --warmup 0 --cpu-rays 0 --no-extras --streams 1 --width 400 --height 400 --no-kernel-timing}
mkdir -p gpurun_out/pmc
# PMC_ONLY="FETCH_SIZE WRITE_SIZE" restricts the passes to the traffic counters; PMC_PROG (default bench.py) is the program
PROG=${PMC_PROG:-bench.py}
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  if [ -n "${PMC_ONLY:-}" ] && ! echo " $PMC_ONLY " | grep -q " $(echo $pass | cut -d' ' -f1) "; then continue; fi
  name=$(echo $pass | cut -d' ' -f1)
  timeout 900 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc_$name -o pmc -- python3 $PROG $ARGS > gpurun_out/pmc/$name.log 2>&1
  echo "pass $name rc=$?"
  python3 tools/pmc_summarize.py /tmp/pmc_$name gpurun_out/pmc/$name.csv
done
