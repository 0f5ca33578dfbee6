#!/usr/bin/env python
"""The reference's command line over its whole test path, twice: `main.py --render_only` on 200 synthetic 800x800 poses (PNG writing
on: 200 x 1.9 MB through the writer threads), the two `rgbs.npy` stacks compared bit for bit, 8 frames spread over the path compared
with the CPU oracle on 2,500 strided rays each, the PNGs decoded back and compared with to8b of the stack.
    python tools/cli_soak.py [n_frames]          (through gpurun; ~1.5 minutes; SOAK_WEIGHTS=trained_like: the trained-like student, i.e. the
    split rungs of `--precision auto` and their watch over the whole path)"""
import os
import subprocess
import sys
import tempfile
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import _pkg  # noqa: E402

_pkg.load()
from efficient_nerf_amd import frontend as fe  # noqa: E402
from oracle import r2l_oracle as O  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
H = 800
d = tempfile.mkdtemp(prefix='r2l_soak_')
if os.environ.get('SOAK_WEIGHTS') == 'trained_like':      # the committed trained-like student instead of the synthetic weights
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'trained_like', 'student_w256d88.npz'))
    sd = {k: torch.from_numpy(z[k]) for k in z.files}
else:
    sd = O.make_r2l_state(seed=0)
ck = os.path.join(d, 'r2l.tar')
fe.save_checkpoint(ck, sd)
stacks = []
for rep in range(2):
    out = os.path.join(d, 'out%d' % rep)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'main.py'), '--model_name', 'R2L', '--config', 'configs/lego_noview_800x800.txt',
                        '--n_sample_per_ray', '16', '--netwidth', '256', '--netdepth', '88', '--use_residual', '--trial.ON', '--trial.body_arch',
                        'resmlp', '--pretrained_ckpt', ck, '--render_only', '--synthetic_poses', str(n), '--H', str(H), '--outdir', out] +
                       (['--watch_every', os.environ['SOAK_WATCH_EVERY']] if 'SOAK_WATCH_EVERY' in os.environ else []),
                       cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    print('\n'.join(ln for ln in r.stdout.splitlines() if ln.startswith(('[precision]', 'Render loop', 'Rendered'))), flush=True)
    stacks.append(np.load(os.path.join(out, 'rgbs.npy')))
same = np.array_equal(stacks[0], stacks[1])
print('two runs of %d frames: rgbs.npy %s' % (n, 'bit-identical' if same else 'DIFFERENT'))
focal = O.focal_from_angle(H)
poses = O.novel_poses(n)
idx = torch.arange(0, H * H, H * H // 2500)[:2500]
worst = 0.0
for i in range(0, n, max(1, n // 8)):
    pts = O.sample_test(O.camera_dirs(H, H, focal), O.sampler_z_vals(16, 2., 6.), poses[i][:3, :4])
    ref = O.r2l_forward(sd, O.positional_embed(pts[idx], 10)).numpy()
    worst = max(worst, float(np.abs(stacks[0][i].reshape(-1, 3)[idx.numpy()] - ref).max()))
print('L_inf against the CPU oracle on 2,500 rays of each of 8 frames: %.2e' % worst)


def read_png(path):
    data = open(path, 'rb').read()
    i, raw = 8, b''
    while i < len(data):
        ln = int.from_bytes(data[i:i + 4], 'big')
        if data[i + 4:i + 8] == b'IDAT':
            raw += data[i + 8:i + 8 + ln]
        i += 12 + ln
    px = np.frombuffer(zlib.decompress(raw), dtype=np.uint8).reshape(H, 1 + H * 3)
    return px[:, 1:].reshape(H, H, 3)


png_ok = all(np.array_equal(read_png(os.path.join(d, 'out0', '%03d.png' % i)), fe.to8b(stacks[0][i])) for i in (0, n // 2, n - 1))
print('PNGs of frames 0, %d, %d decode to to8b of the stack: %s' % (n // 2, n - 1, png_ok))
import shutil  # noqa: E402
shutil.rmtree(d, ignore_errors=True)
sys.exit(0 if same and worst <= 1e-4 and png_ok else 1)
