#!/usr/bin/env python
"""CPU study (no GPU): what arithmetic does the FINE pass of a trained teacher need?

The coarse pass steers sample_pdf (discontinuous: tools/teacher_whole_frame.py) and stays in three fp16 passes.  The fine pass is 75 % of
the points.  Here it is evaluated AT THE SAMPLE POSITIONS OF THE fp32 ORACLE (so that no sample moves) with emulated arithmetic and
composited; L_inf of rgb / acc / depth against the float64 evaluation at the same positions, beside the fp32 oracle's own distance:
  x3        hi(W) hi(a) + hi(W) lo(a) + lo(W) hi(a), fp32 sums                                      (fp16x3_asm: 3.0 pass-equivalents)
  bf6 fixed fp16 pass + bf6(W - hi W) bf6(a) + bf6(W) bf6(a - hi a), activation exponents FIXED at 3 / -9 in the x16 domain
            (what the shipped fp16_fp8 chain does: isa.py ACT_EXP / RES_EXP -- |a| > 14 clamps)     (1.5)
  bf6 cal   the same with per-layer exponents calibrated to the largest |activation| of the layer's input (<= 16 of bf6's 28)   (1.5)
  e4m3 cal  both terms in e4m3, per-layer exponents                                                   (2.0)
  split k   the first k trunk layers (and the embedding k-steps, always) in x3, the rest in `bf6 cal` / `e4m3 cal`
Embedding k-steps (L0, the skip of L5, the view embedding of V) are three fp16 passes in every chain (nerf_gen.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from oracle import r2l_oracle as O  # noqa: E402
from oracle import whole_frame as WF  # noqa: E402

hi = lambda t: t.half().float()


def q_bf6(x):
    a = x.abs().clamp(max=28.0)
    e = torch.floor(torch.log2(a.clamp(min=1e-30))).clamp(min=-2.0)
    step = torch.exp2(e - 2.0)
    return torch.sign(x) * torch.minimum(torch.round(a / step) * step, torch.tensor(28.0))


def q_e4m3(x):
    a = x.abs().clamp(max=448.0)
    e = torch.floor(torch.log2(a.clamp(min=1e-30))).clamp(min=-6.0)
    step = torch.exp2(e - 3.0)
    return torch.sign(x) * torch.minimum(torch.round(a / step) * step, torch.tensor(448.0))


def lin_x3(h, W, b):
    hh, Wh = hi(h), hi(W)
    return F.linear(hh, Wh) + F.linear(h - hh, Wh) + F.linear(hh, hi(W - Wh)) + b


def make_lin_terms(q, top, cal):
    """fp16 pass + two low-precision terms; activations in the x16 domain as in the kernels"""
    def lin(h, W, b, Ea=None):
        a = 16.0 * h
        ah, Wh = hi(a), hi(W)
        e = int(np.frexp(float(W.abs().max()))[1])
        if q is q_bf6:
            el, ew = e - 16, e - 4
        else:
            el, ew = e - 20, e - 8
        if cal:
            m = float(a.abs().max())
            Ea = int(np.ceil(np.log2(max(m, 1e-20) / (16.0 if q is q_bf6 else 256.0))))
        else:
            Ea = 3
        Er = Ea - 12
        t1 = F.linear(q(a / 2.0 ** Ea) * 2.0 ** Ea, q((W - Wh) / 2.0 ** el) * 2.0 ** el)
        t2 = F.linear(q((a - ah) / 2.0 ** Er) * 2.0 ** Er, q(W / 2.0 ** ew) * 2.0 ** ew)
        return (F.linear(ah, Wh) + t1 + t2) / 16.0 + b
    return lin


def make_lin_half(which, cal=False):
    """2.25 pass-equivalents: two fp16 passes + ONE bf6 term.  which = 'w': weights exact (hi + lo fragments), the activation residual in
    bf6; 'a': activations exact (hi + lo), the weight residual in bf6"""
    def lin(h, W, b):
        a = 16.0 * h
        ah, Wh = hi(a), hi(W)
        e = int(np.frexp(float(W.abs().max()))[1])
        el, ew = e - 16, e - 4
        Ea = int(np.ceil(np.log2(max(float(a.abs().max()), 1e-20) / 16.0))) if cal else 3
        Er = Ea - 12
        if which == 'w':
            y = F.linear(ah, Wh) + F.linear(ah, hi(W - Wh)) + F.linear(q_bf6((a - ah) / 2.0 ** Er) * 2.0 ** Er, q_bf6(W / 2.0 ** ew) * 2.0 ** ew)
        else:
            y = F.linear(ah, Wh) + F.linear(hi(a - ah), Wh) + F.linear(q_bf6(a / 2.0 ** Ea) * 2.0 ** Ea, q_bf6((W - Wh) / 2.0 ** el) * 2.0 ** el)
        return y / 16.0 + b
    return lin


def forward(lin_main, k_x3=0, stats=None, lin_lead=None, x3_set=None):
    """teacher_forward with `lin_main` on the 256- / 128-wide sources of the layers behind the first k_x3 trunk layers; embedding
    columns always x3"""
    def f(sd, x, input_ch=63, skips=(4,), dtype=torch.float32):
        W = lambda n: sd[n + '.weight']
        Bv = lambda n: sd[n + '.bias']
        pts, views = x[..., :input_ch], x[..., input_ch:]
        order = [0]

        def main(h, Wm, li):
            if stats is not None:
                stats[li] = max(stats.get(li, 0.), float(h.abs().max()))
            exact = (li in x3_set) if x3_set is not None else li < k_x3
            return ((lin_lead or lin_x3) if exact else lin_main)(h, Wm, torch.zeros(()))
        h = F.relu(lin_x3(pts, W('pts_linears.0'), Bv('pts_linears.0')))
        for i in range(1, 8):
            Wi = W(f'pts_linears.{i}')
            if i == 5:
                y = lin_x3(pts, Wi[:, :input_ch], Bv(f'pts_linears.{i}')) + main(h, Wi[:, input_ch:], i)
            else:
                y = main(h, Wi, i) + Bv(f'pts_linears.{i}')
            h = F.relu(y)
        fa = main(h, torch.cat([W('feature_linear'), W('alpha_linear')], 0), 8) + torch.cat([Bv('feature_linear'), Bv('alpha_linear')])
        feature, alpha = fa[..., :256], fa[..., 256:]
        Wv = W('views_linears.0')
        hv = F.relu(main(feature, Wv[:, :256], 9) + lin_x3(views, Wv[:, 256:], Bv('views_linears.0')))
        rgb = main(hv, W('rgb_linear'), 10) + Bv('rgb_linear')
        return torch.cat([rgb, alpha], -1)
    return f


def main():
    """DEVICE=cuda (the GPU box: torch's own fp32 / fp64 GEMMs as the emulator's arithmetic) evaluates whole frames; CPU: N_RAYS spread"""
    torch.set_num_threads(int(os.environ.get('THREADS', 4)))
    dev = torch.device(os.environ.get('DEVICE', 'cpu'))
    n = int(os.environ.get('N_RAYS', 3000 if dev.type == 'cpu' else WF.H * WF.H))
    chunk = int(os.environ.get('CHUNK', 2048 if dev.type == 'cpu' else 8192))
    sds = tuple({k: v.to(dev) for k, v in sd.items()} for sd in WF.load_teacher())
    H = WF.H
    exact = O.teacher_forward
    cands = [('x3', lin_x3, 0), ('bf6 fixed', make_lin_terms(q_bf6, 28., False), 0), ('bf6 cal', make_lin_terms(q_bf6, 28., True), 0),
             ('e4m3 cal', make_lin_terms(q_e4m3, 448., True), 0)]
    for k in (2, 3, 5):
        cands.append((f'split {k} + bf6 cal', make_lin_terms(q_bf6, 28., True), k))
        cands.append((f'split {k} + e4m3 cal', make_lin_terms(q_e4m3, 448., True), k))
    for k in (3, 4):          # what a hybrid chain WITHOUT calibrated exponents would do (isa.py ACT_EXP / RES_EXP as shipped)
        cands.append((f'split {k} + bf6 fixed', make_lin_terms(q_bf6, 28., False), k))
    fixed = make_lin_terms(q_bf6, 28., False)
    lead = {}
    for k in (3, 4):          # the leading layers with two fp16 passes + one bf6 term instead of three passes (2.25 pass-equivalents)
        for w in 'wa':
            tag = f'split {k} ({"W" if w == "w" else "a"} exact) + bf6 fixed'
            cands.append((tag, fixed, k))
            lead[tag] = make_lin_half(w)
    # ... and the shipped mixed chain on the REBALANCED fine network (teacher.rebalanced_state: the exact power-of-two reparametrisation that
    # brings every bf6 layer's input into the fixed exponents' range -- calibration without a kernel change)
    import _pkg
    _pkg.load()
    from efficient_nerf_amd.teacher import rebalanced_state
    MX = {'h0': 2.8, 'h1': 2.0, 'h2': 2.5, 'h3': 3.0, 'h4': 3.0, 'h5': 5.7, 'h6': 7.9, 'h7': 36.6, 'feature': 52.7, 'views': 147.8}    # largest over the three poses (first run)
    sd_reb = {k: v.to(dev) for k, v in rebalanced_state({k: v.cpu() for k, v in sds[1].items()}, MX)[0].items()}
    for k in (3,):
        cands.append((f'split {k} + bf6 fixed, rebalanced', fixed, k))
    # ... and three-pass layers elsewhere than in front (layer numbers: 1-7 trunk, 8 feature | alpha, 9 views, 10 rgb)
    sets = {}
    for name, ls in (('L1 L2 FA', (1, 2, 8)), ('L1 L2 L7 FA', (1, 2, 7, 8)), ('L1 L2 FA V RGB', (1, 2, 8, 9, 10)), ('L1 FA', (1, 8)), ('L1 L2 L3 FA', (1, 2, 3, 8)),
                     ('L1 L2 RGB', (1, 2, 10)), ('L1 L2 V', (1, 2, 9))):
        tag = f'x3 in {name} + bf6 fixed'
        cands.append((tag, fixed, 0))
        sets[tag] = set(ls)
    only = os.environ.get('STUDY_ONLY')          # comma list of substrings: those candidates only (x3 always)
    if only:
        cands = [c for c in cands if c[0] == 'x3' or any(o in c[0] for o in only.split(','))]
    for pi in range(3):
        ro_all, rd_all = WF.frame_rays(pi)
        idx = torch.arange(0, H * H, max(1, H * H // n))[:n]
        ro_all, rd_all = ro_all[idx].to(dev), rd_all[idx].to(dev)
        acc = {tag: dict(f64=[], x3=[], sig=0.) for tag, _, _ in cands}
        ref32 = []
        stats = {}
        with torch.no_grad():
            for s0 in range(0, ro_all.shape[0], chunk):
                ro, rd = ro_all[s0:s0 + chunk], rd_all[s0:s0 + chunk]
                O.teacher_forward = exact
                o32 = O.render_rays_taps(sds[0], sds[1], ro, rd)
                o64 = O.render_rays_taps(sds[0], sds[1], ro, rd, dtype=torch.float64, z_samples=o32['z_samples'])
                ref32.append((o32['rgb_map'].double() - o64['rgb_map']).abs().max(-1)[0])
                z_all = o32['z_vals']
                pts = ro[..., None, :] + rd[..., None, :] * z_all[..., :, None]
                vd = rd / torch.norm(rd, dim=-1, keepdim=True)
                x3rgb = None
                for tag, lin, k in cands:
                    O.teacher_forward = forward(lin, k, stats if tag in ('bf6 cal', 'x3') else None, lead.get(tag), sets.get(tag))
                    raw = O.run_network(sd_reb if 'rebalanced' in tag else sds[1], pts, vd, netchunk=1 << 22)
                    rgb = O.raw2outputs(raw, z_all, rd, True)[0]
                    if tag == 'x3':
                        x3rgb = rgb
                    acc[tag]['f64'].append((rgb.double() - o64['rgb_map']).abs().max(-1)[0])
                    acc[tag]['x3'].append((rgb - x3rgb).abs().max(-1)[0])
                    sg = o64['raw'][..., 3]
                    acc[tag]['sig'] = max(acc[tag]['sig'], float(((raw[..., 3].double() - sg).abs() / sg.abs().clamp(min=1.))[sg > 1].max()) if (sg > 1).any() else 0.)
        O.teacher_forward = exact
        e32 = torch.cat(ref32)
        print(f'pose {pi}: {ro_all.shape[0]} rays on {dev}; fp32 (torch) vs float64 at the same sample positions: rgb L_inf {e32.max():.2e}, rays > 5e-5: '
              f'{(e32 > 5e-5).sum().item()}; max|a| per main-layer input (L1..L7, FA, V, RGB): {[round(stats[i], 1) for i in sorted(stats)]}', flush=True)
        for tag, _, _ in cands:
            a, b = torch.cat(acc[tag]['f64']), torch.cat(acc[tag]['x3'])
            print(f'  {tag:34s} vs float64: L_inf {a.max():.2e}, rays > 5e-5: {(a > 5e-5).sum().item():4d}, > 1e-4: {(a > 1e-4).sum().item():3d} | vs x3: L_inf '
                  f'{b.max():.2e}, rays > 5e-5: {(b > 5e-5).sum().item():4d}, > 1e-4: {(b > 1e-4).sum().item():3d} | sigma rel (sigma > 1) {acc[tag]["sig"]:.1e}', flush=True)


if __name__ == '__main__':
    main()
