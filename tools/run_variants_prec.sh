# like tools/run_variants.sh for another precision: BT_PREC=fp16_e4m3 VARIANTS="a b" bash tools/run_variants_prec.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for rep in 1 2; do for v in default $VARIANTS; do
  if [ $v = default ]; then python $R/tools/body_time.py; else R2L_LIB_PATH=$R/build_variants/libr2l_$v.so python $R/tools/body_time.py; fi
done; done 2>&1 | grep -v amdgpu.ids >> $R/gpurun_out/variants_prec.log
tail -n $(( 2 * ( $(echo $VARIANTS | wc -w) + 1 ) )) $R/gpurun_out/variants_prec.log
