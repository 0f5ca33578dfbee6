#!/bin/bash
# the mixed rung with K = 2 (shipped), 3, 4 leading three-pass trunk layers on three trained teachers, same box (tools/build_mix_variant.sh builds the
# variant libraries here; tools/sharp_teacher.py measures): auto's probe difference, ms per frame with both exits, whole-frame L_inf against three passes
mkdir -p gpurun_out
OUT=gpurun_out/r06_mix_k_study.txt
: > $OUT
for K in 2 3 4; do
  for T in tests/golden/trained_like scratch/sharp_v2 scratch/sharp_v0_t12k; do
    if [ $K = 2 ]; then unset R2L_LIB_PATH; else export R2L_LIB_PATH=$PWD/build_variants/libr2l_mix$K.so; fi
    echo "## K = $K, teacher $T" >> $OUT
    timeout -k 10 200 python tools/sharp_teacher.py --dir $T 2>&1 | grep "^auto\|^fp16_mix" | cut -c1-420 >> $OUT || exit 1
  done
done
cat $OUT
