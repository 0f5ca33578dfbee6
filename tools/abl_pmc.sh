#!/bin/bash
# usage: tools/abl_pmc.sh <lib.so or ""> <tag> : PMC summary of the fused kernel for one build
LIBP=$1; TAG=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/prof
[ -n "$LIBP" ] && export R2L_LIB_PATH=$LIBP
S_REPS=4 timeout -k 5 100 rocprofv3 --pmc ${PMC_SET:-SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE} --output-format csv -d gpurun_out/prof/abl_$TAG -- python tools/stress.py > gpurun_out/prof/abl_$TAG.log 2>&1
echo "== $TAG"; python tools/pmc_summary.py gpurun_out/prof/abl_$TAG/*/*counter_collection.csv
