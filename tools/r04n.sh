cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04n
python tools/mem_probe.py --hidden 128 --rays 196608 2>/dev/null | tail -1
python tools/mem_probe.py --hidden 128 --rays 640000 2>/dev/null | tail -1
