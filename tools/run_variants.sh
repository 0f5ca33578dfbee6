cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2; do for v in default burst c14 c38 lead2; do
  if [ $v = default ]; then python $R/tools/body_time.py; else R2L_LIB_PATH=$R/build_variants/libr2l_$v.so python $R/tools/body_time.py; fi
done; done 2>&1 | grep -v amdgpu.ids > $R/gpurun_out/variants1.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_body1 -- python $R/tools/body_time.py > $R/gpurun_out/pmc_body1.log 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_body1/*/*counter_collection.csv r2l_body >> $R/gpurun_out/variants1.log 2>&1
cat $R/gpurun_out/variants1.log
