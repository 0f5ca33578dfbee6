#!/bin/bash
# tools/build_teacher_variant.sh NAME [nerf_gen.py options...]: builds build_variants/libr2l_NAME.so with a differently
# generated teacher layer chain (A-B timing; select it with R2L_LIB_PATH).
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
d=$root/build_variants/$name
mkdir -p $d
cp $root/efficient-nerf_amd/csrc/*.hip $root/efficient-nerf_amd/csrc/*.h $root/efficient-nerf_amd/csrc/*.inc $d/
python3 $root/efficient-nerf_amd/csrc/gen/nerf_gen.py --emit $d "$@" | tail -1   # NERF_GEN_FMT=f16 / f16c3 in the environment selects which chain the options regenerate
sed -i 's#"../../include/r2l_hip.h"#"'$root'/include/r2l_hip.h"#' $d/*.hip
cd $d
for f in r2l_kernels r2l_body r2l_capi r2l_comm nerf_capi np_shuffle r2l_generic; do cp $root/efficient-nerf_amd/csrc/$f.o $f.o; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -c nerf_kernels.hip -o nerf_kernels.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/build_variants/libr2l_$name.so r2l_kernels.o r2l_body.o r2l_capi.o r2l_comm.o nerf_kernels.o nerf_capi.o np_shuffle.o r2l_generic.o -ldl
rm -rf $d
echo built build_variants/libr2l_$name.so
