cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04u
(AB_UNIT=mlp AB_CMD="python tools/bench_linear.py --rows 8000000 --shapes 84x128,128x128,128x6,64x64" bash tools/ab_unit.sh "-DRSDF_NOP" "-DRSDF_LIN_NT_STORE"
AB_UNIT=mlp_layer_bwd AB_CMD="python tools/bench_linear.py --rows 8000000 --shapes 84x128,128x128" bash tools/ab_unit.sh "-DRSDF_NOP" "-DRSDF_LIN_NT_STORE") 2>&1 | tee gpurun_out/r04u/ab.log
