#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench step, plus the un-profiled bench line.
# Usage on the GPU box: bash tools/profile_round.sh <name>   -> gpurun_out/prof_<name>/
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
N=${1:-round}
mkdir -p gpurun_out/prof_$N
python3 bench.py > gpurun_out/prof_$N/bench.json 2> gpurun_out/prof_$N/bench.err
# kernel_stats.csv: the step issued on ONE stream (kernels alone on the GPU: what bench.py's roofline object is measured on);
# kernel_stats_two_streams.csv: the shipped two-stream issue order (durations include the neighbour chunk's share)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$N -o prof -- python3 bench.py --steps 1 --warmup 1 --cpu-rays 0 --no-extras --streams 1 > gpurun_out/prof_$N/bench_under_rocprof.json 2> gpurun_out/prof_$N/rocprof.err
find /tmp/prof_$N -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_$N/kernel_stats.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2_$N -o prof -- python3 bench.py --steps 1 --warmup 1 --cpu-rays 0 --no-extras > gpurun_out/prof_$N/bench_under_rocprof_two_streams.json 2>> gpurun_out/prof_$N/rocprof.err
find /tmp/prof2_$N -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_$N/kernel_stats_two_streams.csv \;
tail -1 gpurun_out/prof_$N/bench.json | cut -c1-600
head -12 gpurun_out/prof_$N/kernel_stats.csv | cut -c1-200
