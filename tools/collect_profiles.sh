# Collects the per-round profile set on the GPU box (run through gpurun): bench line, rocprofv3 kernel-trace stats of the
# same command, PMC passes (each in its own run), a determinism stress under --pmc.  Output: gpurun_out/prof_<tag>/.
# usage: bash tools/collect_profiles.sh r02
set -x
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# PART=A: bench lines, kernel trace, the body kernel's PMC passes, traffic.json; PART=B: the other bodies, stress, teacher, power
# (one gpurun call each: the whole set exceeds a call's 20 minutes since bench.py carries the create_data leg)
PART=${PART:-AB}
if [[ $PART == *A* ]]; then
python $R/bench.py > $O/bench_n1_fp16_fp8.json 2> $O/bench.err || exit 1
python $R/bench.py --precision fp16x3 --no-cpu-baseline --no-teacher --no-trained-like > $O/bench_n1_fp16x3.json 2>> $O/bench.err || exit 1
python $R/bench.py --precision fp16_e4m3 --no-teacher --no-trained-like > $O/bench_n1_fp16_e4m3.json 2>> $O/bench.err || exit 1
python $R/bench.py --precision fp16x3_asm --no-teacher --no-trained-like > $O/bench_n1_fp16x3_asm.json 2>> $O/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --no-cpu-baseline --no-teacher --no-create-data --no-trained-like > $O/trace.log 2>&1 || exit 1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F16" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVE_CYCLES" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $O/pmc_$name -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-teacher --no-create-data --no-trained-like > $O/pmc_$name.log 2>&1 || exit 1
  python $R/tools/pmc_summary.py $O/pmc_$name/*/*counter_collection.csv r2l_body_kernel > $O/pmc_$name.txt 2>&1
  python $R/tools/pmc_summary.py $O/pmc_$name/*/*counter_collection.csv 'r2l_head' >> $O/pmc_$name.txt 2>&1
done
# round 6: the kernel trace of the TRAINED-LIKE legs (student on its split rung: r2l_head_kernel<true>, r2l_bodyx_kernel,
# r2l_body8_kernel; teacher: coarse nerf_chain_kernel<false, 2, true>, fine nerf_chain_kernel<false, 2, false, true>), the program directly after `--`
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_trained -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-teacher --no-create-data > $O/trace_trained.log 2>&1 || exit 1
python $R/tools/traffic_json.py $O/pmc_FETCH_SIZE/*/*counter_collection.csv $O/pmc_WRITE_SIZE/*/*counter_collection.csv r2l_body_kernel fp16_fp8 $O/traffic.json
cat $O/bench_n1_fp16_fp8.json
fi
if [[ $PART == *B* ]]; then
# the e4m3 body (fp16_e4m3): wave cycles / MFMA busy / clock for the comparison with the bf6 body
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $O/pmc8_$name -- python $R/bench.py --precision fp16_e4m3 --steps 4 --warmup 1 --no-cpu-baseline --no-teacher --no-create-data --no-trained-like > $O/pmc8_$name.log 2>&1 || exit 1
  python $R/tools/pmc_summary.py $O/pmc8_$name/*/*counter_collection.csv r2l_body8_kernel > $O/pmc8_$name.txt 2>&1
done
# the three-fp16-pass body (fp16x3_asm)
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $O/pmcx_$name -- python $R/bench.py --precision fp16x3_asm --steps 4 --warmup 1 --no-cpu-baseline --no-teacher --no-create-data --no-trained-like > $O/pmcx_$name.log 2>&1 || exit 1
  python $R/tools/pmc_summary.py $O/pmcx_$name/*/*counter_collection.csv r2l_bodyx_kernel > $O/pmcx_$name.txt 2>&1
done
S_PREC=mix S_REPS=20 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VMEM --output-format csv -d $O/pmc_stress -- python $R/tools/stress.py > $O/pmc_stress.log 2>&1
grep -h "stress\|MISMATCH" $O/pmc_stress.log > $O/pmc_stress.txt
# teacher (fp16_fp8): kernel trace + two PMC passes of one 400x400 frame
export T_PREC=fp16_fp8 T_REP=2
python $R/tools/bench_teacher.py > $O/teacher_time.txt 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/teacher_trace -- python $R/tools/bench_teacher.py > $O/teacher_trace.log 2>&1 || exit 1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $O/teacher_pmc_$name -- python $R/tools/bench_teacher.py > $O/teacher_pmc_$name.log 2>&1 || exit 1
  python $R/tools/pmc_summary.py $O/teacher_pmc_$name/*/*counter_collection.csv nerf_chain > $O/teacher_pmc_$name.txt 2>&1
done
# package power and clock while the R2L frame loop runs
BT_FRAMES=2500 bash $R/tools/power_sample.sh > $O/power.txt 2>&1
fi
