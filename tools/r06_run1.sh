#!/bin/bash
# round 6, first GPU call: whole-frame parity of the trained-like teacher, the fine-pass arithmetic study on whole frames (torch on the
# device as the emulator), the new watch tests
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 900 python tools/teacher_whole_frame.py --threads 16 > gpurun_out/r06_whole_frame.log 2>&1 &&
DEVICE=cuda timeout -k 10 600 python tools/teacher_mixed_study.py > gpurun_out/r06_mixed_study.txt 2>&1 &&
timeout -k 10 900 python -m pytest tests/test_split_gpu.py tests/test_trained_like_gpu.py tests/test_teacher_watch_gpu.py -x -q -m gpu -s > gpurun_out/r06_tests1.log 2>&1
echo "exit $?"
tail -5 gpurun_out/r06_whole_frame.log; tail -3 gpurun_out/r06_tests1.log
