#!/bin/bash
# round 6: more points of the trained-like family through `auto` -- tools/train_like.py with other student lengths,
# learning rates and the second scene; reports only (weights are not committed).  usage: bash tools/r06_family.sh NAME:ARGS ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
FIX=$R/tests/golden/trained_like
mkdir -p $R/gpurun_out/r06_family
for spec in "$@"; do
  name=${spec%%:*}; args=${spec#*:}
  out=/tmp/family_$name
  rm -rf $out; mkdir -p $out
  timeout -k 10 1100 python $R/tools/train_like.py --out $out ${args//@FIX/$FIX} > $R/gpurun_out/r06_family/$name.log 2>&1 || { echo "$name failed"; tail -5 $R/gpurun_out/r06_family/$name.log; exit 1; }
  cp $out/report.json $R/gpurun_out/r06_family/$name.json
  grep -h "fitted in\|teacher auto\|\[measure\] student" $R/gpurun_out/r06_family/$name.log | cut -c1-300
done
