#!/bin/bash
# SQ counter passes of a config[2] step (300 x 300 view, 16384-ray chunks), fp32 and 16-bit radiance networks -> pipes.json:
#   bash tools/pmc_c2.sh <out dir>
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=${1:-gpurun_out/pmc_c2}; mkdir -p $O
for P in fp32 bf16; do
  timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_c2_$P -o pmc -- python3 tools/bench_c2.py --width 300 --height 300 --chunk 16384 --steps 1 --tex-precision $P > $O/sq_$P.log 2>&1
  python3 tools/pmc_summarize.py /tmp/pmc_c2_$P $O/sq_$P.csv
  python3 tools/pmc_pipes.py $O/sq_$P.csv $O/c2_pipes_$P.json
done
python3 - $O <<'PY'
import json, sys
for p in ("fp32", "bf16"):
    d = json.load(open(f"{sys.argv[1]}/c2_pipes_{p}.json"))
    for k, v in d.items():
        if "pair_kernel" in k:
            print(p, k[:40], {a: round(b, 3) for a, b in v.items() if isinstance(b, float) and a != "valu_instructions"})
PY
