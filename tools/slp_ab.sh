# same-box A/B of the build flag -fno-slp-vectorize (product) against the SLP-vectorised build (build_variants/libr2l_slp.so, made by
# `VARIANT_FLAGS=" " bash tools/build_variant.sh slp`): the frame in fp16_fp8 (head launch = HIP ray code around the asm), the
# compiler-scheduled fp16x3 kernel, the teacher frame (chain kernel's HIP prologue, scan kernels).  Run through gpurun.
R=$GRAFT_REPO_ROOT
cd /tmp
for rep in 1 2; do
for lib in "" $R/build_variants/libr2l_slp.so; do
  export R2L_LIB_PATH=$lib; [ -z "$lib" ] && unset R2L_LIB_PATH
  echo "== ${lib:-product (no packed fp32)}"
  BT_FRAMES=60 BT_PREC=fp16_fp8 python $R/tools/body_time.py 2>&1 | grep frame
  BT_FRAMES=40 BT_PREC=fp16x3 python $R/tools/body_time.py 2>&1 | grep frame
  T_PREC=fp16_fp8 T_REP=6 python $R/tools/bench_teacher.py 2>&1 | grep teacher
done
done
