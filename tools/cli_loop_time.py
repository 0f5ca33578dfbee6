#!/usr/bin/env python
"""is `python main.py --render_only` the loop bench.py times?  Saves the synthetic W256D88 checkpoint, runs the
reference's command line on 20 synthetic 800x800 test poses (PNG writing on: the writer threads run beside the loop) and prints
the CLI's own lines (per-frame time, `Render loop: ... rays/s`), then bench.py's value on the same box for comparison.
    python tools/cli_loop_time.py [n_frames]          (through gpurun)"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

_pkg.load()
from efficient_nerf_amd import frontend as fe  # noqa: E402
from oracle import r2l_oracle as O  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
fpb = sys.argv[2] if len(sys.argv) > 2 else '0'       # --frames_per_batch
d = tempfile.mkdtemp(prefix='r2l_cli_')
ck = os.path.join(d, 'r2l.tar')
fe.save_checkpoint(ck, O.make_r2l_state(seed=0))
cmd = [sys.executable, os.path.join(ROOT, 'main.py'), '--model_name', 'R2L', '--config', 'configs/lego_noview_800x800.txt',
       '--n_sample_per_ray', '16', '--netwidth', '256', '--netdepth', '88', '--use_residual', '--trial.ON', '--trial.body_arch', 'resmlp',
       '--pretrained_ckpt', ck, '--render_only', '--synthetic_poses', str(n), '--H', '800', '--outdir', os.path.join(d, 'out'),
       '--frames_per_batch', fpb]
r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True)
lines = r.stdout.splitlines()
keep = [ln for ln in lines if ln.startswith(('[precision]', 'Render loop', 'Rendered'))]
times = [float(ln.split('time for this frame: ')[1].rstrip('s')) for ln in lines if 'time for this frame' in ln]
print('$ main.py --model_name R2L --config configs/lego_noview_800x800.txt ... --render_only --synthetic_poses %d --frames_per_batch %s   (rc %d)' % (n, fpb, r.returncode))
for ln in keep:
    print(ln)
if times:
    ts = sorted(times[1:] or times)
    print('per-frame lines: %d; median %.4f s, min %.4f s (first frame %.4f s)' % (len(times), ts[len(ts) // 2], ts[0], times[0]))
if r.returncode:
    print(r.stderr[-2000:])
b = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-teacher',
                    '--no-create-data'], cwd=ROOT, capture_output=True, text=True)
try:
    j = json.loads([ln for ln in b.stdout.splitlines() if ln.startswith('{')][-1])
    print('bench.py --steps 20 on the same box: %.3e rays/s, %.3f ms/step (median %.3f)' % (j['value'], j['ms_per_step'], j['median_ms']))
    for ln in keep:
        if ln.startswith('Render loop'):
            v = float(ln.split(' = ')[1].split(' rays/s')[0])
            print('CLI render loop / bench.py = %.3f' % (v / j['value']))
except Exception as e:  # noqa: BLE001
    print('bench.py failed:', e, b.stderr[-1500:])
