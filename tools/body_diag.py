"""Diagnostics for the body kernel: error pattern by row tile / wave / column tile, run-to-run differences."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np, torch
import _pkg; _pkg.load()
from efficient_nerf_amd import R2LEngine, PREC_FP16_FP8
from oracle import r2l_oracle as O
from body_check import image_to_rays, rays_to_image

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sd = O.make_r2l_state(seed=0, netdepth=2 + 2 * nb)
mode = sys.argv[3] if len(sys.argv) > 3 else 'normal'
if mode == 'w16':      # weights exactly representable in fp16: the (w - hi) term is identically zero
    for k in sd:
        if k.startswith('body') and k.endswith('weight'): sd[k] = sd[k].half().float()
if mode == 'w8':       # weights exactly representable in e4m3-ish (few bits): both w and hi exact, a-residual term exact in w
    for k in sd:
        if k.startswith('body') and k.endswith('weight'): sd[k] = (sd[k] * 64).round() / 64
print('mode', mode)
eng = R2LEngine(64, 64, O.focal_from_angle(64), n_block=nb, precision=PREC_FP16_FP8).load_state_dict(sd)
g = torch.Generator().manual_seed(1)
xr = torch.relu(torch.randn(nt * 128, 256, generator=g)); S = 16.0
xin = rays_to_image(xr * S).cuda()
outs = [eng.debug_body(xin).cpu() for _ in range(4)]
x = xr.double(); Bsum = torch.zeros(256, dtype=torch.float64)
def f16(a): return a.half().double()
xh = xr.double()
for i in range(nb):
    W1, b1 = sd[f'body.{i}.body.0.weight'].double(), sd[f'body.{i}.body.0.bias'].double()
    W2, b2 = sd[f'body.{i}.body.2.weight'].double(), sd[f'body.{i}.body.2.bias'].double()
    x = x + torch.relu(x @ W1.T + b1) @ W2.T + b2
    hh = torch.relu(f16(xh) @ f16(W1).T + b1); xh = xh + f16(hh) @ f16(W2).T + b2
    Bsum += b2
print('fp16x1 model err', (xh - x).abs().max().item())
for k, o in enumerate(outs):
    got = image_to_rays(o).double() / S + Bsum
    e = (got - x).abs()
    print(f'run {k}: L_inf {e.max().item():.3e} mean {e.mean().item():.3e} frac>5e-5 {(e > 5e-5).double().mean().item():.4f}')
e = (image_to_rays(outs[0]).double() / S + Bsum - x).abs()
er = e.reshape(nt, 4, 2, 16, 16, 16)   # tile, wave, ray half, ray16, 16-feature group, f16
print('by row tile u :', ' '.join(f'{v:.1e}' for v in er.amax(dim=(0, 1, 2, 3, 5)).tolist()))
print('by wave       :', ' '.join(f'{v:.1e}' for v in er.amax(dim=(0, 2, 3, 4, 5)).tolist()))
print('by col tile   :', ' '.join(f'{v:.1e}' for v in er.amax(dim=(0, 1, 3, 4, 5)).tolist()))
print('by ray in tile:', ' '.join(f'{v:.1e}' for v in er.amax(dim=(0, 1, 2, 4, 5)).tolist()))
print('by feat in tile:', ' '.join(f'{v:.1e}' for v in er.amax(dim=(0, 1, 2, 3, 4)).tolist()))
d = (outs[0] - outs[1]).abs()
print('run0 vs run1: n differing', int((d > 0).sum()), 'of', d.numel(), 'max', d.max().item() / S)
dd = (image_to_rays(outs[0]) - image_to_rays(outs[1])).abs().reshape(nt, 4, 2, 16, 16, 16)
print('diff by row tile u:', ' '.join(f'{v:.1e}' for v in (dd.amax(dim=(0, 1, 2, 3, 5)) / S).tolist()))
print('diff by wave      :', ' '.join(f'{v:.1e}' for v in (dd.amax(dim=(0, 2, 3, 4, 5)) / S).tolist()))
