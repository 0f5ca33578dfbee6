# tools/get_rays_probe (the library's get_rays kernel in its SLP-vectorised form, standalone process) beside another process's
# nerf_chain_kernel: reproduces the wrong d.x of profiles/r04_gpu_sharing.txt and says which operand was wrong
R=$GRAFT_REPO_ROOT
cd $R
GS_HEAVY_ONLY=1 GS_HEAVY_SECONDS=${HEAVY_S:-25} python tools/gpu_sharing_check.py ${HEAVY_MODE:-c} > /tmp/heavy2.log 2>&1 &
HP=$!
sleep 12
tools/get_rays_probe ${REPS:-1500}
wait $HP
