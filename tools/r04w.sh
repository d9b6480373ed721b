cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04w
run() { for i in 1 2 3; do python tools/bench_step.py --steps 40 --settle 80 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('  ms/step', round(d['ms_per_step'],2), 'rsdf', d['rsdf_kernel_ms_per_step'])"; done; }
echo "== shipped"; run
AB_UNIT=hashgrid_fd7 AB_TAIL=3 AB_CMD="bash -c run" bash -c 'true'
cd rise_sdf_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -I../../include -DRSDF_Q_PLAIN_STORE -DRSDF_X2_PLAIN_IMAGE -DRSDF_R_WGS=768 -c hashgrid_fd7.hip -o /tmp/h.o && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls _build/*.o | grep -v "/hashgrid_fd7.o") /tmp/h.o -o /tmp/libvar.so && cd ../..
echo "== plain stores, 768 reducer workgroups"; export RSDF_LIB=/tmp/libvar.so; run
