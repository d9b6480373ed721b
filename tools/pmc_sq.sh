#!/bin/bash
# one SQ counter pass over bench.py (see tools/pmc_passes.sh)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format csv -d /tmp/pmc_sq -o pmc -- python3 bench.py --steps 1 --warmup 0 --cpu-rays 0 --no-extras --streams 1 --width 400 --height 400 --no-kernel-timing > gpurun_out/pmc/sq.log 2>&1
python3 tools/pmc_summarize.py /tmp/pmc_sq gpurun_out/pmc/sq.csv
timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmc_sq2 -o pmc -- python3 bench.py --steps 1 --warmup 0 --cpu-rays 0 --no-extras --streams 1 --width 400 --height 400 --no-kernel-timing > gpurun_out/pmc/sq2.log 2>&1
python3 tools/pmc_summarize.py /tmp/pmc_sq2 gpurun_out/pmc/sq2.csv
