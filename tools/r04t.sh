cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04t
(AB_SRC=mlp_x2 bash tools/ab_x2.sh "" "-DRSDF_NOP -I../../include" "-DRSDF_X2_NT_LOAD -I../../include" "-DRSDF_X2_NT_DMA -I../../include" "-DRSDF_X2_NT_LOAD -DRSDF_X2_NT_DMA -I../../include") 2>&1 | tee gpurun_out/r04t/ab2.log
