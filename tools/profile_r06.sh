#!/bin/bash
# Round-6 profile set on one box: the driver's exact bench command, rocprofv3 kernel stats of the same workload on one stream,
# SQ counter passes of the shipped x2 / hash kernels (matrix- and vector-pipe busy), the FETCH / WRITE passes -> pmc summary,
# kernel stats of a config[2] step and of the training step.
#   bash tools/profile_r06.sh <name> <build tag>   -> gpurun_out/prof_<name>/
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
N=${1:-r06}; TAG=${2:-unknown}
O=gpurun_out/prof_$N; mkdir -p $O/pmc
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
tail -1 $O/bench_driver_cmd.json | cut -c1-300
S1="--steps 1 --warmup 1 --cpu-rays 0 --no-extras --streams 1"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$N -o prof -- python3 bench.py $S1 > $O/bench_under_rocprof.json 2> $O/rocprof.err
find /tmp/prof_$N -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
head -8 $O/kernel_stats.csv | cut -c1-160
# SQ passes (400 x 400 view: counter collection serialises dispatches; same kernels, same chunk size)
V="--steps 1 --warmup 0 --cpu-rays 0 --no-extras --streams 1 --width 400 --height 400 --no-kernel-timing"
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_sq_$N -o pmc -- python3 bench.py $V > $O/pmc/sq.log 2>&1
python3 tools/pmc_summarize.py /tmp/pmc_sq_$N $O/pmc/sq.csv
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d /tmp/pmc_sq2_$N -o pmc -- python3 bench.py $V > $O/pmc/sq2.log 2>&1
python3 tools/pmc_summarize.py /tmp/pmc_sq2_$N $O/pmc/sq2.csv
python3 tools/pmc_pipes.py $O/pmc/sq.csv $O/pmc/pipes.json
PMC_ONLY="FETCH_SIZE WRITE_SIZE" bash tools/pmc_passes.sh
cp gpurun_out/pmc/FETCH_SIZE.csv gpurun_out/pmc/WRITE_SIZE.csv $O/pmc/ 2>/dev/null
SAMPLES=$(python3 bench.py $V 2>/dev/null | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['config']['samples_per_step'])")
python3 tools/pmc_summary.py $O/pmc $SAMPLES $O/pmc_summary.json "$TAG"
# config[2] on the full view, one step under the kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2_$N -o prof -- python3 tools/bench_c2.py --width 800 --height 800 --chunk 16384 --steps 1 > $O/c2_under_rocprof.json 2>> $O/rocprof.err
find /tmp/prof_c2_$N -name "*kernel_stats.csv" -exec cp {} $O/c2_kernel_stats.csv \;
head -12 $O/c2_kernel_stats.csv | cut -c1-160
bash tools/step_kernel_stats.sh > $O/c3_step_kernel_stats.txt 2>&1 || true
tail -5 $O/c3_step_kernel_stats.txt
# round 6 additions: the 16-bit radiance mode of config[2] under the kernel trace, the prefilter kernels alone, the RCCL probe
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2h_$N -o prof -- python3 tools/bench_c2.py --width 800 --height 800 --chunk 16384 --steps 1 --tex-precision bf16 > $O/c2_bf16_under_rocprof.json 2>> $O/rocprof.err
find /tmp/prof_c2h_$N -name "*kernel_stats.csv" -exec cp {} $O/c2_bf16_kernel_stats.csv \;
head -8 $O/c2_bf16_kernel_stats.csv | cut -c1-160
python3 tools/bench_prefilter.py > $O/prefilter_alone.json 2>/dev/null; cat $O/prefilter_alone.json
python3 tests/rccl_probe.py > $O/rccl_probe.txt 2>&1; tail -1 $O/rccl_probe.txt | cut -c1-400
