"""A/B on one box: the body kernel's fused tail against the three-launch form (same context, alternating blocks of
frames).  Run through gpurun: python tools/fused_tail_ab.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import r2l_oracle as O  # noqa: E402
import _pkg  # noqa: E402
_pkg.load()
from efficient_nerf_amd import PREC_FP16_FP8, R2LEngine  # noqa: E402

H = 800
focal = O.focal_from_angle(H)
sd = O.make_r2l_state(seed=0)
eng = R2LEngine(H, H, focal, precision=PREC_FP16_FP8).load_state_dict(sd)
poses = torch.stack([torch.as_tensor(O.pose_spherical(t, -30., 4.))[:3, :4].float() for t in range(0, 360, 18)]).cuda()
out = torch.empty((1, H * H, 3), device='cuda')
for i in range(5):
    eng.render_batch(poses[i], out=out)
torch.cuda.synchronize()
res = {0: [], 1: []}
for rep in range(6):
    for mode in (1, 0):
        eng._set_fused_tail(mode)
        eng.render_batch(poses[0], out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20):
            eng.render_batch(poses[i], out=out)
        torch.cuda.synchronize()
        res[mode].append((time.perf_counter() - t0) / 20 * 1e3)
for mode in (1, 0):
    r = sorted(res[mode])
    print('fused' if mode else 'split', 'ms/frame: median %.3f min %.3f  all %s' % (r[len(r) // 2], r[0], ' '.join('%.3f' % x for x in res[mode])))
