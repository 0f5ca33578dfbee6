#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; echo "bench exit $?"
python tools/cli_soak.py 200 > gpurun_out/r06_soak_synth.txt 2>&1; echo "soak exit $?"; grep "Render loop\|watch\|bit-identical\|L_inf" gpurun_out/r06_soak_synth.txt | cut -c1-250
SOAK_WEIGHTS=trained_like python tools/cli_soak.py 200 > gpurun_out/r06_soak_trained.txt 2>&1; echo "soak exit $?"; grep "Render loop\|watch\|bit-identical\|L_inf" gpurun_out/r06_soak_trained.txt | cut -c1-250
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | cut -c1-400
