#!/bin/bash
# tools/build_variant.sh NAME [body_gen.py options...]: builds build_variants/libr2l_NAME.so with a differently
# generated body loop (diagnostics / A-B timing; select it with R2L_LIB_PATH).
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
d=$root/build_variants/$name
mkdir -p $d
cp $root/efficient-nerf_amd/csrc/*.hip $root/efficient-nerf_amd/csrc/*.h $root/efficient-nerf_amd/csrc/*.inc $d/
# `--fmt fp8` / `--fmt f16` variants replace the includes of r2l_body8_kernel / r2l_bodyx_kernel (time them with BT_PREC)
stem=r2l_body
case " $* " in *" --fmt fp8 "*) stem=r2l_body8;; *" --fmt f16 "*) stem=r2l_bodyx;; esac
python3 $root/efficient-nerf_amd/csrc/gen/body_gen.py --emit $d/${stem}_asm.inc "$@" > /dev/null || exit 1
[ $stem = r2l_bodyx ] || python3 $root/efficient-nerf_amd/csrc/gen/body_gen.py --guard --emit $d/${stem}_guard_asm.inc "$@" > /dev/null || exit 1
# CAPI_DEF="-DR2L_SLICE_TILES=1280": extra definitions for the host side
# HEAD_OPTS="--dma-gap 2 ...": a differently generated head layer as well (gen/head_gen.py options)
if [ -n "$HEAD_OPTS" ]; then python3 $root/efficient-nerf_amd/csrc/gen/head_gen.py --emit $d $HEAD_OPTS > /dev/null || exit 1; fi
sed -i 's#"../../include/r2l_hip.h"#"'$root'/include/r2l_hip.h"#' $d/*.hip
cd $d
# a variant generated with `--fmt bf6r` streams 22 KiB chunks: its packer and LDS size are selected at compile time
DEF=""
case " $* " in *" --fmt bf6r "*) DEF="-DR2L_BF6R_STREAM";; esac
# VARIANT_FLAGS replaces the product's -fno-slp-vectorize (csrc/Makefile) and recompiles every file: VARIANT_FLAGS=" " builds the
# library WITH the SLP vectorizer's packed-fp32 ops (the A/B of profiles/r04_gpu_sharing.txt)
CF=${VARIANT_FLAGS--fno-slp-vectorize}
for f in r2l_kernels r2l_body r2l_capi r2l_comm nerf_kernels nerf_capi np_shuffle; do
  if [ -n "${VARIANT_FLAGS+x}" ] || [ $f = r2l_body ] || [ $f = r2l_kernels -a -n "$HEAD_OPTS" ] || [ $f = r2l_capi -a -n "$DEF$CAPI_DEF" ] || [ ! -f $root/efficient-nerf_amd/csrc/$f.o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off $CF $DEF $CAPI_DEF -c $f.hip -o $f.o
  else cp $root/efficient-nerf_amd/csrc/$f.o $f.o; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/build_variants/libr2l_$name.so r2l_kernels.o r2l_body.o r2l_capi.o r2l_comm.o nerf_kernels.o nerf_capi.o np_shuffle.o -ldl
rm -rf $d
echo built build_variants/libr2l_$name.so
