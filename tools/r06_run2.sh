#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 600 python tools/teacher_whole_frame.py --threads 16 > gpurun_out/r06_whole_frame.log 2>&1 &&
timeout -k 10 900 python -m pytest tests/test_trained_like_gpu.py tests/test_split_gpu.py tests/test_teacher_watch_gpu.py -x -q -m gpu -s > gpurun_out/r06_tests2.log 2>&1
echo "tests exit $?"; tail -3 gpurun_out/r06_tests2.log
bash tools/r06_family.sh "v0_lr0.4:--teacher-from @FIX --lr-scale 0.4" "v0_lr2:--teacher-from @FIX --lr-scale 2.0"
