"""CPU study: RGB L_inf of the R2L W256D88 network when the two correction terms of the fp16 main pass use cheaper
operand formats (OCP fp8, MX fp6 / fp4, static per-layer power-of-two scales or per-32-block scales on the weights).
float64 arithmetic with exact products: only the operand quantisation is modelled.
    python tools/quant_study.py [n_rays]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import r2l_oracle as O


def fmt_values(ebits, mbits, bias, vmax=None):
    vals = {0.0}
    for e in range(2 ** ebits):
        for m in range(2 ** mbits):
            vals.add(m / 2 ** mbits * 2.0 ** (1 - bias) if e == 0 else (1 + m / 2 ** mbits) * 2.0 ** (e - bias))
    v = np.array(sorted(vals))
    return v[v <= vmax] if vmax else v


FMT = {'e4m3': fmt_values(4, 3, 7, 448), 'e5m2': fmt_values(5, 2, 15, 57344), 'e2m3': fmt_values(2, 3, 1),
       'e3m2': fmt_values(3, 2, 3), 'e2m1': fmt_values(2, 1, 1)}
TOP = {k: int(np.floor(np.log2(v[-1]))) for k, v in FMT.items()}


def quant(x, fmt, trunc=False):
    v = FMT[fmt]
    a = np.abs(x)
    idx = np.searchsorted(v, a).clip(1, len(v) - 1)
    lo, hi = v[idx - 1], v[idx]
    q = np.where(a >= hi, hi, lo) if trunc else np.where(a - lo <= hi - a, lo, hi)
    return np.sign(x) * np.minimum(q, v[-1])


def qs(x, fmt, e, trunc=False):   # quantise x / 2^e, e scalar or array
    return quant(x * 2.0 ** (-np.asarray(e, dtype=np.float64)), fmt, trunc) * 2.0 ** np.asarray(e, dtype=np.float64)


def mx_exp(w, fmt):               # per block of 32 inputs: block max -> top binade of fmt
    m = np.abs(w).reshape(w.shape[0], -1, 32).max(-1)
    return np.repeat(np.floor(np.log2(np.maximum(m, 1e-300))) - TOP[fmt], 32, axis=1)


def run(sd, emb, corr, nb=43):
    f16 = lambda a: a.astype(np.float16).astype(np.float64)
    g = lambda k: sd[k].double().numpy()
    h0 = np.maximum(emb @ g('head.0.weight').T + g('head.0.bias'), 0)
    x = h0.copy()

    def layer(a, W, b):
        y = f16(a) @ f16(W).T + b
        return y if corr is None else y + corr(a, W)
    for i in range(nb):
        h = np.maximum(layer(x, g(f'body.{i}.body.0.weight'), g(f'body.{i}.body.0.bias')), 0)
        x = x + layer(h, g(f'body.{i}.body.2.weight'), g(f'body.{i}.body.2.bias'))
    y = (x + h0) @ g('tail.0.weight').T + g('tail.0.bias')
    return 1 / (1 + np.exp(-y))


def make_corr(fa, fw, mx=False, S=16.0, da=0, dal=0, dwl=0, dw=0, trunc_a=False):
    f16 = lambda a: a.astype(np.float16).astype(np.float64)

    def corr(a, W):
        ex = int(np.frexp(np.abs(W).max())[1])        # max|w| in [2^(ex-1), 2^ex)
        wh = f16(W); ah = f16(a)
        wl, al = W - wh, a - ah
        qa = qs(a * S, fa, 6 - TOP[fa] + da, trunc_a) / S            # scaled activations < 2^7
        qal = qs(al * S, fa, -5 - TOP[fa] + dal) / S                 # their fp16 residuals < 2^-5
        if mx:
            qwl = qs(wl, fw, mx_exp(wl, fw)); qw = qs(W, fw, mx_exp(W, fw))
        else:
            qwl = qs(wl, fw, ex - 13 - TOP[fw] + dwl); qw = qs(W, fw, ex - 1 - TOP[fw] + dw)
        return qa @ qwl.T + qal @ qw.T
    return corr


def stress_state(kind, seed=21):
    """weight distributions of tools/weight_stress.py"""
    sd = O.make_r2l_state(seed=seed, netdepth=88)
    g = torch.Generator().manual_seed(5)
    for k, w in sd.items():
        if not k.endswith('weight'):
            continue
        std = w.std()
        if kind == 'laplace':
            u = torch.rand(w.shape, generator=g) - 0.5
            sd[k] = (-torch.sign(u) * torch.log1p(-2 * u.abs()) * std / np.sqrt(2)).float()
        elif kind == 'sparse':
            sd[k] = w * (torch.rand(w.shape, generator=g) < 0.5).float() * np.sqrt(2)
        elif kind == 'outlier':
            w2 = w.clone()
            idx = torch.randint(0, w.numel(), (w.numel() // 2000,), generator=g)
            w2.view(-1)[idx] *= 12.0
            sd[k] = w2
    return sd


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    if len(sys.argv) > 2:   # stress distributions, fixed candidate schemes
        H = 64
        pts = O.sample_test(O.camera_dirs(H, H, O.focal_from_angle(H)), O.sampler_z_vals(16, 2., 6.),
                            torch.as_tensor(O.pose_spherical(30., -30., 4.))[:3, :4])
        emb = O.positional_embed(pts[torch.linspace(0, pts.shape[0] - 1, n).long()]).double().numpy()
        for kind in ('uniform', 'laplace', 'sparse', 'outlier'):
            sd = stress_state(kind)
            f16 = lambda a: a.astype(np.float16).astype(np.float64)
            exact = run(sd, emb, lambda a, W: a @ W.T - f16(a) @ f16(W).T)
            r = {'fp16x1': np.abs(run(sd, emb, None) - exact).max(),
                 'e5m2 x e4m3': np.abs(run(sd, emb, make_corr('e5m2', 'e4m3', da=TOP['e5m2'] - 6, dal=TOP['e5m2'] + 5)) - exact).max(),
                 'e3m2 x e2m3 static': np.abs(run(sd, emb, make_corr('e3m2', 'e2m3', da=0, dal=-1, dwl=1, dw=1)) - exact).max(),
                 'e3m2 x e3m2 static': np.abs(run(sd, emb, make_corr('e3m2', 'e3m2', da=1, dal=0, dwl=1, dw=1)) - exact).max(),
                 'e3m2 x e2m3 MX': np.abs(run(sd, emb, make_corr('e3m2', 'e2m3', True, da=0, dal=-1)) - exact).max()}
            print(kind, {k: f'{v:.2e}' for k, v in r.items()}, flush=True)
        return
    sd = O.make_r2l_state(seed=0, netdepth=88)
    H = 64
    focal = O.focal_from_angle(H)
    c2w = O.pose_spherical(30., -30., 4.)
    pts = O.sample_test(O.camera_dirs(H, H, focal), O.sampler_z_vals(16, 2., 6.), torch.as_tensor(c2w)[:3, :4])
    idx = torch.linspace(0, pts.shape[0] - 1, n).long()
    emb = O.positional_embed(pts[idx]).double().numpy()
    exact = run(sd, emb, lambda a, W: a @ W.T - a.astype(np.float16).astype(np.float64) @ W.astype(np.float16).astype(np.float64).T)
    print(f'{"fp16x1 (no correction)":44s} L_inf {np.abs(run(sd, emb, None) - exact).max():.3e}')
    rows = [('a:e5m2 w:e4m3 (shipping)', make_corr('e5m2', 'e4m3', da=TOP['e5m2'] - 6 + 0, dal=TOP['e5m2'] + 5, dwl=0, dw=0))]
    for fa, fw in (('e3m2', 'e2m3'), ('e3m2', 'e3m2'), ('e2m3', 'e2m3'), ('e2m3', 'e3m2'), ('e2m1', 'e2m1')):
        for mx in (False, True):
            best = None
            for da in (0, 1) if not mx else (0, 1):
                for dal in (0, -1, -2):
                    for dw in ((0, 1) if not mx else (0,)):
                        c = make_corr(fa, fw, mx, da=da, dal=dal, dwl=dw, dw=dw)
                        e = np.abs(run(sd, emb, c) - exact).max()
                        if best is None or e < best[0]:
                            best = (e, da, dal, dw)
            print(f'{"a:" + fa + " w:" + fw + (" MX weights" if mx else " static"):44s} L_inf {best[0]:.3e}  (da {best[1]} dal {best[2]} dw {best[3]})', flush=True)
    for name, c in rows:
        print(f'{name:44s} L_inf {np.abs(run(sd, emb, c) - exact).max():.3e}')
    print(f'{"a:e5m2 truncated (top byte of fp16), w:e4m3":44s} L_inf '
          f'{np.abs(run(sd, emb, make_corr("e5m2", "e4m3", da=TOP["e5m2"] - 6, dal=TOP["e5m2"] + 5, trunc_a=True)) - exact).max():.3e}')


if __name__ == '__main__':
    main()
