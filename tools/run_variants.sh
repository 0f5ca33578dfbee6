# times build variants of the body kernel (tools/build_variant.sh) back to back on one box, twice
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for rep in 1 2; do for v in default $VARIANTS; do
  if [ $v = default ]; then python $R/tools/body_time.py; else R2L_LIB_PATH=$R/build_variants/libr2l_$v.so python $R/tools/body_time.py; fi
done; done 2>&1 | grep -v amdgpu.ids > $R/gpurun_out/variants.log
cat $R/gpurun_out/variants.log
