"""Experiment: frames alternate between two contexts on two streams (each context owns its x image), so the head launch of
frame i+1 can fill the CUs that frame i's last, partly empty round of ray tiles leaves idle (5,000 tiles on 256 CUs =
19.5 rounds).  Prints ms per frame for one stream and for two.  Run through gpurun: python tools/two_streams_ab.py"""
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
engs = [R2LEngine(H, H, focal, precision=PREC_FP16_FP8).load_state_dict(sd) for _ in range(2)]
poses = torch.stack([torch.as_tensor(O.pose_spherical(t, -30., 4.))[:3, :4].float() for t in range(0, 360, 9)]).cuda()
outs = [torch.empty((1, H * H, 3), device='cuda') for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
for e, o in zip(engs, outs):
    for i in range(3):
        e.render_batch(poses[i], out=o)
torch.cuda.synchronize()
ref = engs[0].render_batch(poses[7]).clone()


def run(n_streams, frames=40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(frames):
        k = i % n_streams
        with torch.cuda.stream(streams[k]):
            engs[k].render_batch(poses[i % 40], out=outs[k])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / frames * 1e3


for rep in range(3):
    a = run(1)
    b = run(2)
    print('one stream %.3f ms/frame   two streams %.3f ms/frame' % (a, b))
with torch.cuda.stream(streams[1]):
    engs[1].render_batch(poses[7], out=outs[1])
torch.cuda.synchronize()
print('same frame from both contexts bitwise equal:', bool(torch.equal(outs[1], ref)))
