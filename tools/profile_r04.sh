#!/bin/bash
# Round-4 profile set on one box: the driver's exact bench command, rocprofv3 kernel stats (one stream / two streams), the
# FETCH / WRITE PMC passes -> pmc summary, the generic gather's L2 counters, the training step's kernel stats.
#   bash tools/profile_r04.sh <name> <build tag>   -> gpurun_out/prof_<name>/
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
N=${1:-r04}; TAG=${2:-unknown}
O=gpurun_out/prof_$N; mkdir -p $O/pmc
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
tail -1 $O/bench_driver_cmd.json | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$N -o prof -- python3 bench.py --steps 1 --warmup 1 --cpu-rays 0 --no-extras --streams 1 > $O/bench_under_rocprof.json 2> $O/rocprof.err
find /tmp/prof_$N -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2_$N -o prof -- python3 bench.py --steps 1 --warmup 1 --cpu-rays 0 --no-extras > $O/bench_under_rocprof_two_streams.json 2>> $O/rocprof.err
find /tmp/prof2_$N -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_two_streams.csv \;
head -8 $O/kernel_stats.csv | cut -c1-160
PMC_ONLY="FETCH_SIZE WRITE_SIZE" bash tools/pmc_passes.sh
cp gpurun_out/pmc/FETCH_SIZE.csv gpurun_out/pmc/WRITE_SIZE.csv $O/pmc/ 2>/dev/null
SAMPLES=$(python3 bench.py --steps 1 --warmup 0 --cpu-rays 0 --no-extras --streams 1 --width 400 --height 400 --no-kernel-timing 2>/dev/null | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['config']['samples_per_step'])")
python3 tools/pmc_summary.py $O/pmc $SAMPLES $O/pmc_summary.json "$TAG"
bash tools/pmc_gather.sh; cp gpurun_out/pmc/gather_l2.csv $O/pmc/ 2>/dev/null
bash tools/step_kernel_stats.sh > $O/c3_step_kernel_stats.txt 2>&1 || true
tail -5 $O/c3_step_kernel_stats.txt
