cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_x2.py -q 2>&1 | tail -3
AB_SRC=hashgrid_fd7 bash tools/ab_x2.sh "" "-DRSDF_X2_WAVES=3" "-DRSDF_X2_WAVES=4" "-DRSDF_X2_WAVES=3 -DRSDF_FWD_GROUP=2048" "-DRSDF_X2_WAVES=3 -DRSDF_FWD_GROUP=8192"
