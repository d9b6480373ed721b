cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04l
AB_UNIT=mlp AB_CMD="python tools/bench_linear.py --rows 8000000" bash tools/ab_unit.sh "-DRSDF_LIN_KSU=4 -DRSDF_BWDW_OCC=4" "-DRSDF_LIN_KSU=4 -DRSDF_BWDW_OCC=4 -DRSDF_LIN_KSU_BI=2" "-DRSDF_LIN_KSU=8 -DRSDF_BWDW_OCC=4 -DRSDF_LIN_KSU_BI=4" 2>&1 | tee gpurun_out/r04l/ab2.log
