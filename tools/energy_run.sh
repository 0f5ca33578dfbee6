# One box, one call: the energy probe (W per instruction class), then package power / sclk while the two generated kernels
# run back to back (R2L body: tools/body_time.py frame loop; teacher chain: tools/bench_teacher.py), so that
# tools/energy_account.py prices the kernels with energies measured under the same power cap.  Run through gpurun.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/r04_energy
mkdir -p $OUT
timeout -k 10 420 $R/tools/energy_probe > $OUT/probe.txt 2>&1 || exit 1
EP_AB=1 timeout -k 10 200 $R/tools/energy_probe > $OUT/probe_ab.txt 2>&1 || exit 1
sample() {   # $1 = label, $2 = pid: six samples while the loop runs
  sleep 10
  for i in 1 2 3 4 5 6; do echo -n "$1 "; rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk" | tr '\n' ' '; echo; sleep 0.7; done
}
rocm-smi --showmaxpower 2>&1 | grep -i "max" > $OUT/power.txt
BT_FRAMES=2500 BT_PREC=fp16_fp8 python $R/tools/body_time.py > $OUT/body.txt 2>&1 &
PID=$!
sample body >> $OUT/power.txt
wait $PID || exit 1
T_PREC=fp16_fp8 T_REP=330 python $R/tools/bench_teacher.py > $OUT/teacher.txt 2>&1 &
PID=$!
sample chain >> $OUT/power.txt
wait $PID || exit 1
cat $OUT/power.txt $OUT/body.txt $OUT/teacher.txt
