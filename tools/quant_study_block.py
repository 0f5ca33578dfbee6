"""CPU study: what per-(ray, 32-element K block) power-of-two scales of the ACTIVATION operands of the two bf6 correction terms
-- the block scaling v_mfma_scale_f32_32x32x64_f8f6f4 takes per lane, computed on the fly from each lane's own 32 values --
would buy over one calibrated scale per operand set (what the kernels do), on tools/range_sweep.py's networks.  float64
arithmetic with exact products: only the operand quantisation is modelled.   python tools/quant_study_block.py [n_rays]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from oracle import r2l_oracle as O
from quant_study import TOP, qs
from quant_study_gain import S, f16, cal_exp, make_corr, run


def block_exp(x, fmt, bs=32):
    """per (row, block of bs columns): E with blockmax / 2^E in (2^(TOP-1), 2^TOP]"""
    m = np.abs(x).reshape(x.shape[0], -1, bs).max(-1)
    fr, e = np.frexp(m / 2.0 ** TOP[fmt])
    e = np.where(fr == 0.5, e - 1, e)
    e = np.where(m == 0, -40, e)
    return np.repeat(np.maximum(e, -40), bs, axis=1)


def qs_block(x, fmt, E):
    out = np.empty_like(x)
    for ev in np.unique(E):
        m = E == ev
        out[m] = qs(x[m], fmt, int(ev))
    return out


def make_corr_block(fa, fal, fwl, fw, wblock=False):
    def corr(a, W):
        wh, ah = f16(W), f16(a)
        wl, al = W - wh, a - ah
        Ea = block_exp(a * S, fa)
        qa = qs_block(a * S, fa, Ea) / S
        qal = qs_block(al * S, fal, Ea - 12 + (TOP[fa] - TOP[fal])) / S
        if wblock:
            Ew = block_exp(W, fw)
            qwl = qs_block(wl, fwl, Ew - 12 + (TOP[fw] - TOP[fwl]))
            qw = qs_block(W, fw, Ew)
        else:
            ex = int(np.frexp(np.abs(W).max())[1])
            qwl = qs(wl, fwl, ex - 12 - TOP[fwl])
            qw = qs(W, fw, ex - TOP[fw])
        return qa @ qwl.T + qal @ qw.T
    return corr


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    H = 64
    focal = O.focal_from_angle(H)
    b6, e4 = ('e3m2',) * 4, ('e4m3',) * 4
    schemes = [('bf6 set scale', make_corr(*b6)), ('bf6 block a', make_corr_block(*b6)), ('bf6 block a+w', make_corr_block(*b6, wblock=True)),
               ('e4m3 set scale', make_corr(*e4)), ('e4m3 block a', make_corr_block(*e4))]
    for seed in (0, 1):
        for gain in (1.0, 1.1, 1.2, 1.3, 1.4, 1.5):
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
                errs.append([np.abs(run(sd, emb, c)[0] - exact).max() for _, c in schemes])
            e = np.max(np.array(errs), 0)
            print('seed %d gain %.2f E %d: ' % (seed, gain, top - 4) + '  '.join('%s %.2e' % (nm, v) for (nm, _), v in zip(schemes, e)),
                  flush=True)


if __name__ == '__main__':
    main()
