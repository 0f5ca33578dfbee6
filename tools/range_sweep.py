"""How the fp16_fp8 error grows with the magnitude of the residual stream: W256D88 networks with every body weight
scaled by a gain, 200x200 frames; prints the largest measured activation exponent (R2LEngine.act_exponents: the set's
|a| * 16 <= 16 * 2^E, i.e. |a| <= 2^E) and L_inf of fp16_fp8 against fp16x3 (itself 6e-7 from the fp32 oracle).
Basis of the threshold of `--precision auto` (frontend.py).  Run through gpurun: python tools/range_sweep.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import _pkg
_pkg.load()
from efficient_nerf_amd import R2LEngine, PREC_FP16X3, PREC_FP16_FP8
from oracle import r2l_oracle as O

H = 200
focal = O.focal_from_angle(H)
for seed in (0, 1):
    for gain in (1.0, 1.05, 1.1, 1.15, 1.2, 1.25, 1.3, 1.4):
        sd = O.make_r2l_state(seed=seed)
        for k in sd:
            if 'body' in k and k.endswith('weight'):
                sd[k] = sd[k] * gain
        worst = 0.0
        e3 = R2LEngine(H, H, focal, precision=PREC_FP16X3).load_state_dict(sd)
        e8 = R2LEngine(H, H, focal, precision=PREC_FP16_FP8).load_state_dict(sd)
        for th in (0., 120., 240.):
            c2w = O.pose_spherical(th, -30., 4.)
            a, b = e3.render(c2w).cpu(), e8.render(c2w).cpu()
            worst = max(worst, (a - b).abs().max().item())
        ex = e8.act_exponents()
        print('seed %d gain %.2f: max exponent x %2d  h %2d   L_inf(fp16_fp8 - fp16x3) = %.2e   finite %s'
              % (seed, gain, max(ex[0::2]), max(ex[1::2]), worst, bool(torch.isfinite(b).all())), flush=True)
        e3.close()
        e8.close()
