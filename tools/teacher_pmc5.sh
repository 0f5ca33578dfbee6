# round 5: PMC + kernel-trace passes of teacher frames per mode (run through gpurun): bash tools/teacher_pmc5.sh <tag> <T_PREC> <kernel pattern>
set -x
TAG=${1:-r05}
export T_PREC=${2:-fp16x1} T_REP=2
PAT=${3:-nerf_chain}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/teacher_${TAG}_${T_PREC}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/tools/bench_teacher.py > $O/time.txt 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/tools/bench_teacher.py > $O/trace.log 2>&1 || exit 1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $O/pmc_$name -- python $R/tools/bench_teacher.py > $O/pmc_$name.log 2>&1 || exit 1
  python $R/tools/pmc_summary.py $O/pmc_$name/*/*counter_collection.csv $PAT > $O/pmc_$name.txt 2>&1
done
cat $O/time.txt $O/pmc_*.txt
cp $O/trace/*/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
head -6 $O/kernel_stats.csv
