# times build variants of the teacher chain (tools/build_teacher_variant.sh) back to back on one box, twice
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export T_PREC=${T_PREC:-fp16_fp8} T_REP=3
for rep in 1 2; do for v in default $VARIANTS; do
  if [ $v = default ]; then python $R/tools/bench_teacher.py; else R2L_LIB_PATH=$R/build_variants/libr2l_$v.so python $R/tools/bench_teacher.py; fi 2>&1 | grep -v amdgpu.ids | sed "s/^/$v: /"
done; done > $R/gpurun_out/teacher_variants.log
cat $R/gpurun_out/teacher_variants.log
