#!/bin/bash
# `python bench.py --gpus N` (the driver's command, no torchrun) and the torchrun form rehearsed on the ONE-GPU box: the ranks share the card
# (gloo between them, --allow-fallback), so the rates mean nothing; the run shows the launcher, the ragged row shards, the exponent agreement
# and the assembled frame on every rank.     bash tools/launcher_rehearsal.sh <round>      (through gpurun; ~3 minutes)
R=${1:-r06}
OUT=gpurun_out/${R}_launcher_rehearsal.txt
mkdir -p gpurun_out
export R2L_DIST_BACKEND=gloo
FLAGS="--steps 4 --warmup 2 --no-cpu-baseline --no-teacher --no-create-data --no-trained-like --allow-fallback"
: > $OUT
for N in 2 6; do
  timeout -k 10 300 python bench.py --gpus $N $FLAGS > gpurun_out/reh_$N.json 2> gpurun_out/reh_$N.err
  echo "python bench.py --gpus $N: exit $? $(python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/reh_$N.json').read().strip().splitlines()[-1])
    print('one JSON line, n_gpus', d['n_gpus'], 'precision', d['config'].get('precision'), 'gather_check', d.get('gather_check') or d['config'].get('gather_check'))
except Exception as e:
    print('NO JSON LINE', e)
PY
)" >> $OUT || exit 1
done
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 $FLAGS > gpurun_out/reh_t2.json 2> gpurun_out/reh_t2.err
echo "torchrun --nproc-per-node 2 bench.py --gpus 2: exit $? $(tail -1 gpurun_out/reh_t2.json | cut -c1-200)" >> $OUT
cat $OUT
