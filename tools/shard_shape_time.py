"""Per-rank work of bench.py at N ranks, timed on ONE GPU: rank 0's share of an N-GPU step is N poses x rows
[0, 800/N) = 640,000 rays in one call, whatever N is.  Prints ms per step for N = 1, 2, 4, 8 (no collective: the gather
is the driver's to measure).  Run through gpurun: python tools/shard_shape_time.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import r2l_oracle as O  # noqa: E402
import _pkg  # noqa: E402
_pkg.load()
from efficient_nerf_amd import PREC_FP16_FP8, R2LEngine  # noqa: E402
from efficient_nerf_amd import dist as D  # noqa: E402

H = 800
focal = O.focal_from_angle(H)
eng = R2LEngine(H, H, focal, precision=PREC_FP16_FP8).load_state_dict(O.make_r2l_state(seed=0))
poses = torch.stack([torch.as_tensor(O.pose_spherical(t, -30., 4.))[:3, :4].float() for t in range(0, 360, 9)]).cuda()
for world in (1, 2, 4, 8):
    r0, r1 = D.row_shard(H, 0, world)
    out = torch.empty((world, (r1 - r0) * H, 3), device='cuda')
    for i in range(3):
        eng.render_batch(poses[i * world:(i + 1) * world], rows=(r0, r1), out=out)
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(5):
            eng.render_batch(poses[i * world:(i + 1) * world], rows=(r0, r1), out=out)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 5 * 1e3)
    print('N=%d: rank 0 renders %d poses x rows [%d, %d) = %d rays per step: %.3f ms (min of 3 x 5 steps)'
          % (world, world, r0, r1, world * (r1 - r0) * H, min(ts)))
