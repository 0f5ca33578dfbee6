#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests/test_teacher_mix_gpu.py -x -q -m gpu -s > gpurun_out/r06_mix_tests.log 2>&1
echo "mix tests exit $?"; tail -25 gpurun_out/r06_mix_tests.log | cut -c1-400
for m in fp16x3_asm fp16_mix fp16_fp8; do T_PREC=$m T_REP=5 timeout -k 10 300 python tools/bench_teacher.py 2>&1 | grep teacher; done
timeout -k 10 600 python tools/teacher_whole_frame.py --threads 16 --modes fp16_mix --out gpurun_out/teacher_whole_frame_mix > gpurun_out/r06_whole_frame_mix.log 2>&1; grep "^==" gpurun_out/teacher_whole_frame_mix.txt
