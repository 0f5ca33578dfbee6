#!/usr/bin/env python
"""Development timing (GPU box): `create_data rand` at the reference's own save-group size (utils/create_data.py:812-872:
100 random poses at 400x400, i_save = 100, split_size = 4096 -> 3,906 shards of 147 KB) through
efficient_nerf_amd.create_data.create_rand, beside the bare render loop of the same poses.
    python tools/create_data_time.py [n_pose] [H] [outdir]"""
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import _pkg  # noqa: E402

_pkg.load()
from efficient_nerf_amd import NeRFEngine, PRECISIONS  # noqa: E402
from efficient_nerf_amd import create_data as CD  # noqa: E402
from efficient_nerf_amd.teacher import get_rays  # noqa: E402
from oracle import r2l_oracle as O  # noqa: E402

n_pose = int(sys.argv[1]) if len(sys.argv) > 1 else 100
H = W = int(sys.argv[2]) if len(sys.argv) > 2 else 400
out = sys.argv[3] if len(sys.argv) > 3 else '/tmp/r2l_pseudo'
focal = O.focal_from_angle(W)
eng = NeRFEngine(H, W, focal, precision=PRECISIONS['fp16_fp8']).load_state_dicts(O.make_teacher_state(1), O.make_teacher_state(2))
s = CD.RandStream()
poses = [(s.rand_pose(), focal * s.rand_focal_scale()) for _ in range(n_pose)]
eng.render(poses[0][0][:3, :4])
torch.cuda.synchronize()
t0 = time.perf_counter()
for p, f in poses:
    ro, rd = get_rays(H, W, f, p[:3, :4], device=eng.device)
    eng.render_rays(ro.reshape(-1, 3), rd.reshape(-1, 3))
torch.cuda.synchronize()
t_render = time.perf_counter() - t0
print(f'bare render loop: {n_pose} poses {H}x{W}: {t_render:.2f} s = {n_pose / t_render:.2f} poses/s', flush=True)
shutil.rmtree(out, ignore_errors=True)
t0 = time.perf_counter()
kw = {}
if 'timings' in CD.create_rand.__code__.co_varnames:
    kw['timings'] = tm = {}
n = CD.create_rand(eng, H, W, focal, n_pose, out, i_save=min(100, n_pose), split_size=4096, stream=CD.RandStream(), log=lambda *a: None, **kw)
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f'create_rand: {n} shards, {t_all:.2f} s = {n_pose / t_all:.2f} poses/s; render share {t_render / t_all:.2f}; '
      f'extrapolated --n_pose_kd 10000: {1e4 / n_pose * t_all / 3600:.2f} h', flush=True)
if kw:
    print('timings:', {k: (round(v, 3) if isinstance(v, float) else v) for k, v in tm.items()}, flush=True)
nbytes = sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out))
print(f'{nbytes / 1e6:.0f} MB on disk at {out}', flush=True)
shutil.rmtree(out, ignore_errors=True)
