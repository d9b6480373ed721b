cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s
AB_ARGS="--only bwd --forms pts --rays 28672" timeout 1500 bash tools/ab_hash_prof.sh "-DRSDF_NOP" "-DRSDF_DPL_NT_LOAD" "-DRSDF_R_WGS=1536" "-DRSDF_R_WGS=2048" "-DRSDF_R_WGS=3072" "-DRSDF_Q_PLAIN_STORE" 2>&1 | tee gpurun_out/r04s/ab2.log
