# tools/pk_sgpr_probe beside another process's nerf_chain_kernel (the heavy process starts first: the order that reproduces), then alone.
R=$GRAFT_REPO_ROOT
cd $R
GS_HEAVY_ONLY=1 GS_HEAVY_SECONDS=40 python tools/gpu_sharing_check.py c > /tmp/heavy.log 2>&1 &
HP=$!
sleep 12
echo "== beside another process's nerf_chain_kernel"
tools/pk_sgpr_probe 2500
wait $HP
echo "== alone"
tools/pk_sgpr_probe 2500
echo "== the library's get_rays kernel (SLP build) in a standalone process, alone"
tools/get_rays_probe 500
GS_HEAVY_ONLY=1 GS_HEAVY_SECONDS=30 python tools/gpu_sharing_check.py c > /tmp/heavy2.log 2>&1 &
HP=$!
sleep 12
echo "== ... beside another process's nerf_chain_kernel"
tools/get_rays_probe 2500
wait $HP
