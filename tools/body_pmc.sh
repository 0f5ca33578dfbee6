# two PMC passes of the R2L body kernel on one frame (run through gpurun): bash tools/body_pmc.sh <tag>
TAG=${1:-x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/body_pmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export BT_FRAMES=6
python $R/tools/body_time.py 2>&1 | tail -1 > $O/time.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $O/p1 -- python $R/tools/body_time.py > $O/p1.log 2>&1 || exit 1
python $R/tools/pmc_summary.py $O/p1/*/*counter_collection.csv ${PAT:-r2l_body} > $O/p1.txt 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $O/p2 -- python $R/tools/body_time.py > $O/p2.log 2>&1 || exit 1
python $R/tools/pmc_summary.py $O/p2/*/*counter_collection.csv ${PAT:-r2l_body} > $O/p2.txt 2>&1
cat $O/time.txt $O/p1.txt $O/p2.txt
