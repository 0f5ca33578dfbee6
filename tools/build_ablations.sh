#!/bin/bash
# Development: timing-only ablation builds of the fused kernel (outputs are wrong by design).
set -e
cd "$(dirname "$0")/../efficient-nerf_amd/csrc"
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -shared"
VARS=(${ABLS:-NOEPI})
for v in "${VARS[@]}"; do
  name=$(echo $v | tr -d ' -' | sed 's/DR2L_ABL_//')
  /opt/rocm/bin/hipcc $F -D${PREFIX:-R2L_ABL_}$v r2l_kernels.hip r2l_capi.hip nerf_kernels.hip nerf_capi.hip -o ../abl_$name.so &
done
wait
ls -la ../abl_*.so
