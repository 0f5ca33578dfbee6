#!/bin/bash
# tools/build_mix_variant.sh K: builds build_variants/libr2l_mixK.so whose R2L_PREC_FP16_MIX chain (and its second-exit build) runs trunk layers
# L1 .. LK in three fp16 passes instead of the shipped L1 .. L2 (study: error against time; select it with R2L_LIB_PATH).
set -e
K=$1
root=$(cd "$(dirname "$0")/.." && pwd)
d=$root/build_variants/mix$K
mkdir -p $d
cp $root/efficient-nerf_amd/csrc/*.hip $root/efficient-nerf_amd/csrc/*.h $root/efficient-nerf_amd/csrc/*.inc $d/
bm=$(NERF_GEN_FMT=mix NERF_GEN_MIX_K=$K python3 $root/efficient-nerf_amd/csrc/gen/nerf_gen.py --emit $d | grep -o "stream bytes [0-9]*" | cut -d' ' -f3)
bs=$(NERF_GEN_FMT=mixs NERF_GEN_MIX_K=$K python3 $root/efficient-nerf_amd/csrc/gen/nerf_gen.py --emit $d | grep -o "stream bytes [0-9]*" | cut -d' ' -f3)
echo "mix K=$K: stream bytes $bm, with the second exit $bs"
sed -i "s/#define NERF_MIX_K .*/#define NERF_MIX_K $K/; s/#define NERF_CHAINM_STREAM_BYTES .*/#define NERF_CHAINM_STREAM_BYTES $bm/; s/#define NERF_CHAINMS_STREAM_BYTES .*/#define NERF_CHAINMS_STREAM_BYTES $bs/" $d/nerf_common.h
sed -i 's#"../../include/r2l_hip.h"#"'$root'/include/r2l_hip.h"#' $d/*.hip
cd $d
for f in r2l_kernels r2l_body r2l_capi r2l_comm np_shuffle r2l_generic; do cp $root/efficient-nerf_amd/csrc/$f.o $f.o; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -c nerf_kernels.hip -o nerf_kernels.o &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -c nerf_capi.hip -o nerf_capi.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/build_variants/libr2l_mix$K.so r2l_kernels.o r2l_body.o r2l_capi.o r2l_comm.o nerf_kernels.o nerf_capi.o np_shuffle.o r2l_generic.o -ldl
rm -rf $d
echo built build_variants/libr2l_mix$K.so
