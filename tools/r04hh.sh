cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04hh
(AB_SRC=mlp_x2 bash tools/ab_x2.sh "" "-DRSDF_NOP -I../../include" "-DRSDF_X2_FWD_WAVES=8 -I../../include" "-DRSDF_X2_FWD_WAVES=16 -I../../include"
AB_SRC=hashgrid_fd7 bash tools/ab_x2.sh "" "-DRSDF_X2_WAVES=4 -I../../include" "-DRSDF_STAGE_RECS=6 -I../../include" "-DP_THREADS_CFG=512 -I../../include") 2>&1 | tee gpurun_out/r04hh/ab.log
