"""CPU study for the `auto` precision ladder: RGB L_inf of W256D88 networks whose body weights are scaled by a gain
(tools/range_sweep.py's networks: activation exponents 3 .. 6) when the two correction terms of the fp16 main pass use
different operand formats, every operand set scaled by its own calibrated power of two (what the kernels do).
float64 arithmetic with exact products: only the operand quantisation is modelled.
    python tools/quant_study_gain.py [n_rays]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from oracle import r2l_oracle as O
from quant_study import FMT, TOP, qs

S = 16.0


def f16(a):
    return a.astype(np.float16).astype(np.float64)


def cal_exp(x, fmt):
    """E with max|x| / 2^E in (2^(TOP-1), 2^TOP]: the calibration rule of r2l_calib_finalize_kernel for any format"""
    m = float(np.abs(x).max())
    if m == 0:
        return 0
    fr, e = np.frexp(m / 2.0 ** TOP[fmt])
    return int(e - 1 if fr == 0.5 else e)


def make_corr(fa, fal, fwl, fw):
    def corr(a, W):
        wh, ah = f16(W), f16(a)
        wl, al = W - wh, a - ah
        ea = cal_exp(a * S, fa)
        qa = qs(a * S, fa, ea) / S
        qal = qs(al * S, fal, ea - 12 + (TOP[fa] - TOP[fal])) / S        # residuals: 2^12 finer than the values
        ex = int(np.frexp(np.abs(W).max())[1])
        qwl = qs(wl, fwl, ex - 12 - TOP[fwl])
        qw = qs(W, fw, ex - TOP[fw])
        return qa @ qwl.T + qal @ qw.T
    return corr


def run(sd, emb, corr, nb=43):
    g = lambda k: sd[k].double().numpy()
    h0 = np.maximum(emb @ g('head.0.weight').T + g('head.0.bias'), 0)
    x = h0.copy()
    top = 0

    def layer(a, W, b):
        y = f16(a) @ f16(W).T + b
        return y if corr is None else y + corr(a, W)
    for i in range(nb):
        top = max(top, cal_exp(x * S, 'e3m2'))
        h = np.maximum(layer(x, g(f'body.{i}.body.0.weight'), g(f'body.{i}.body.0.bias')), 0)
        top = max(top, cal_exp(h * S, 'e3m2'))
        x = x + layer(h, g(f'body.{i}.body.2.weight'), g(f'body.{i}.body.2.bias'))
    y = (x + h0) @ g('tail.0.weight').T + g('tail.0.bias')
    return 1 / (1 + np.exp(-y)), top


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    H = 64
    focal = O.focal_from_angle(H)
    schemes = [('bf6 x bf6 (fp16_fp8)', ('e3m2',) * 4), ('a e2m3, w bf6', ('e2m3', 'e2m3', 'e3m2', 'e3m2')),
               ('e2m3 x e2m3', ('e2m3',) * 4), ('a e5m2, w e4m3', ('e5m2', 'e5m2', 'e4m3', 'e4m3')),
               ('e4m3 x e4m3', ('e4m3',) * 4), ('a e4m3, w bf6', ('e4m3', 'e4m3', 'e3m2', 'e3m2'))]
    for seed in (0, 1):
        for gain in (1.0, 1.1, 1.15, 1.2, 1.25, 1.3):
            sd = O.make_r2l_state(seed=seed, netdepth=88)
            for k in sd:
                if 'body' in k and k.endswith('weight'):
                    sd[k] = sd[k] * gain
            errs = []
            for th in (0., 120.):
                c2w = O.pose_spherical(th, -30., 4.)
                pts = O.sample_test(O.camera_dirs(H, H, focal), O.sampler_z_vals(16, 2., 6.), torch.as_tensor(c2w)[:3, :4])
                idx = torch.linspace(0, pts.shape[0] - 1, n).long()
                emb = O.positional_embed(pts[idx]).double().numpy()
                exact, top = run(sd, emb, lambda a, W: a @ W.T - f16(a) @ f16(W).T)
                errs.append([np.abs(run(sd, emb, make_corr(*f))[0] - exact).max() for _, f in schemes])
            e = np.max(np.array(errs), 0)
            print('seed %d gain %.2f E %d: ' % (seed, gain, top - 4) + '  '.join('%s %.2e' % (nm, v) for (nm, _), v in zip(schemes, e)),
                  flush=True)


if __name__ == '__main__':
    main()
