# tools/pk_localise_probe beside another process's nerf_chain_kernel (heavy process first), then alone
R=$GRAFT_REPO_ROOT
cd $R
GS_HEAVY_ONLY=1 GS_HEAVY_SECONDS=${HEAVY_S:-45} python tools/gpu_sharing_check.py ${HEAVY_MODE:-c} > /tmp/heavy3.log 2>&1 &
HP=$!
sleep 12
echo "== beside another process's nerf_chain_kernel"
tools/pk_localise_probe ${REPS:-400}
wait $HP
echo "== alone"
tools/pk_localise_probe ${REPS:-400}
