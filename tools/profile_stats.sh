#!/bin/bash
# The two rocprofv3 --kernel-trace --stats passes of tools/profile_round.sh without the un-profiled bench run (when the
# round's bench.json already exists):  bash tools/profile_stats.sh <name>  -> gpurun_out/prof_<name>/
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
N=${1:-round}
mkdir -p gpurun_out/prof_$N
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$N -o prof -- python3 bench.py --steps 1 --warmup 1 --cpu-rays 0 --no-extras --streams 1 > gpurun_out/prof_$N/bench_under_rocprof.json 2> gpurun_out/prof_$N/rocprof.err
find /tmp/prof_$N -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_$N/kernel_stats.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2_$N -o prof -- python3 bench.py --steps 1 --warmup 1 --cpu-rays 0 --no-extras > gpurun_out/prof_$N/bench_under_rocprof_two_streams.json 2>> gpurun_out/prof_$N/rocprof.err
find /tmp/prof2_$N -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_$N/kernel_stats_two_streams.csv \;
head -8 gpurun_out/prof_$N/kernel_stats.csv | cut -c1-200
