cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04g
for sh in 0 0.02 0.005; do
  for rays in 28672 4096; do
    echo "== shell $sh rays $rays"
    AB_ARGS="--only bwd --forms pts --rays $rays --shell $sh" timeout 300 bash tools/ab_hash_prof.sh "-DRSDF_REC_FP32"
    python tools/bench_hash_fd7.py --forms pts --rays $rays --shell $sh 2>/dev/null | tail -1
  done
done 2>&1 | tee gpurun_out/r04g/shell.log
