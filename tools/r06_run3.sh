#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
DEVICE=cuda timeout -k 10 600 python tools/teacher_mixed_study.py > gpurun_out/r06_mixed_study.txt 2>&1
bash tools/r06_family.sh "v0_s24k:--teacher-from @FIX --student-steps 24000" "v1_lr2:--variant 1 --lr-scale 2.0"
