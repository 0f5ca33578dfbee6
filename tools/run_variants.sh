# times build variants of the body kernel (tools/build_variant.sh) back to back on one box, twice, then one PMC pass
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for rep in 1 2; do for v in default $VARIANTS; do
  if [ $v = default ]; then python $R/tools/body_time.py; else R2L_LIB_PATH=$R/build_variants/libr2l_$v.so python $R/tools/body_time.py; fi
done; done 2>&1 | grep -v amdgpu.ids > $R/gpurun_out/variants.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_body -- python $R/tools/body_time.py > $R/gpurun_out/pmc_body.log 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_body/*/*counter_collection.csv r2l_body >> $R/gpurun_out/variants.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc_body2 -- python $R/tools/body_time.py > $R/gpurun_out/pmc_body2.log 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_body2/*/*counter_collection.csv r2l_body >> $R/gpurun_out/variants.log 2>&1
cat $R/gpurun_out/variants.log
