#!/usr/bin/env python
"""Rate of the generic fp32 layer path (efficient-nerf_amd/generic.py) beside the fused kernels: the README's W256D88 network
through GenericR2L at 800x800, two smaller R2L shapes, and a 4 x 128 / 8 x 256 teacher through GenericNeRF at 400x400."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import _pkg; _pkg.load()
from efficient_nerf_amd.generic import GenericR2L, GenericNeRF
from oracle import r2l_oracle as O

T = dict(body_arch='resmlp', n_block=-1, n_learnable=2, res_scale=1.0, inact='relu', outact='none')
pose = O.novel_poses(1)[0][:3, :4]


def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n


for name, D, W, ns, L in (('W256 D88 (README)', 88, 256, 16, 10), ('W128 D44', 44, 128, 16, 10), ('W64 D12, 8 samples, L 6', 12, 64, 8, 6)):
    H = 800
    eng = GenericR2L(H, H, O.focal_from_angle(H), n_sample=ns, L=L, netdepth=D, netwidth=W, trial=T)
    eng.load_state_dict(O.make_v3_2_state(0, D, W, eng.input_dim, '', 'relu', T))
    dt = timed(lambda: eng.render(pose))
    print(f'GenericR2L {name}: {dt * 1e3:.1f} ms per 800x800 frame = {H * H / dt:.3e} rays/s = {eng.flops_per_ray * H * H / dt / 1e12:.1f} TFLOP/s fp32 '
          f'({eng.flops_per_ray * H * H / dt / 1e12 / 157.3:.2f} of the fp32 MFMA peak)')
for name, D, W in (('8 x 256 (configs/lego.txt)', 8, 256), ('4 x 128', 4, 128)):
    H = 400
    eng = GenericNeRF(H, H, O.focal_from_angle(H), white_bkgd=True, netdepth=D, netwidth=W, netdepth_fine=D, netwidth_fine=W, chunk=1 << 14)
    eng.load_state_dicts(O.make_nerf_state(1, D, W), O.make_nerf_state(2, D, W))
    dt = timed(lambda: eng.render(pose), n=2)
    print(f'GenericNeRF {name}: {dt * 1e3:.0f} ms per 400x400 frame = {H * H / dt:.3e} rays/s = {eng.flops_per_ray * H * H / dt / 1e12:.1f} TFLOP/s fp32')
