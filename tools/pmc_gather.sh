#!/bin/bash
# L2 hit rate and request counts of the generic gather's kernels (tools/bench_gather.py) -- one counter pass, no other tracing
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
rm -rf /tmp/pmc_g
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d /tmp/pmc_g -o pmc -- python3 tools/bench_gather.py > gpurun_out/pmc/gather.log 2>&1
python3 tools/pmc_summarize.py /tmp/pmc_g gpurun_out/pmc/gather_l2.csv
head -4 gpurun_out/pmc/gather_l2.csv
