# samples rocm-smi power / clocks / temperature while the R2L frame loop runs (run through gpurun)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocm-smi --showpower --showclocks --showtemp --showperflevel --showmaxpower 2>&1 | grep -v "^$" | head -40
BT_FRAMES=${BT_FRAMES:-400} BT_PREC=${BT_PREC:-fp16_fp8} python $R/tools/body_time.py > /tmp/bt.log 2>&1 &
PID=$!
sleep 12
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' '; echo; sleep 0.7; done
wait $PID
tail -1 /tmp/bt.log
