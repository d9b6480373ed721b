cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04x
(AB_SRC=hashgrid_fd7 bash tools/ab_x2.sh "" "-DRSDF_NOP -I../../include" "-DRSDF_FWD_GROUP=2048 -I../../include" "-DRSDF_FWD_GROUP=8192 -I../../include" "-DRSDF_FWD_GROUP=16384 -I../../include" "-DRSDF_BWD_GROUP=128 -I../../include" "-DRSDF_BWD_GROUP=256 -I../../include" "-DRSDF_BWD_GROUP=1024 -I../../include" "-DRSDF_X2_WAVES=2 -I../../include") 2>&1 | tee gpurun_out/r04x/ab.log
