#!/bin/bash
# the rebalancing target 8 -> 32: the tests that touch the mixed rung, the whole-frame classification in fp16_mix, the sharp teacher again
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_teacher_mix_gpu.py tests/test_trained_like_gpu.py tests/test_teacher_gpu.py tests/test_graph_capture_gpu.py -x -q -m gpu -s > gpurun_out/r06_t32_tests.log 2>&1
echo tests exit $?; tail -3 gpurun_out/r06_t32_tests.log
timeout -k 10 400 python bench.py > gpurun_out/r06_t32_bench.json 2> gpurun_out/r06_t32_bench.err
echo bench exit $?
mkdir -p gpurun_out/sharp_v2 && cp scratch/sharp_v2/*.npz gpurun_out/sharp_v2/
timeout -k 10 300 python tools/sharp_teacher.py > gpurun_out/sharp_v2/run32.log 2>&1
echo sharp exit $?; grep -v "^\[teacher" gpurun_out/sharp_v2/run32.log | cut -c1-400
