cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04k
AB_ARGS="--only bwd --forms pts --rays 28672" timeout 1500 bash tools/ab_hash_prof.sh "-DRSDF_NOP" "-DRSDF_BWD_GROUP=32" "-DRSDF_BWD_GROUP=128" "-DMERGE_MAX_RUNS=64" "-DMERGE_MAX_RUNS=24" "-DRSDF_STAGE_RECS=3" "-DRSDF_R_UNROLL=16" 2>&1 | tee gpurun_out/r04k/ab.log
