#!/usr/bin/env python
"""The mixed rung's calibration swept: rebalanced_state(keep, target) -- the exact power-of-two reparametrisation of the fine network that
`auto` applies before it measures R2L_PREC_FP16_MIX -- for other targets than the shipped 8.0, on a teacher that misses the rung's 5e-5
(the thin-bar scene, tools/sharp_teacher.py) and on the committed fixture.  Measured: fp16_mix against fp16x3_asm (same coarse pass, so
the same sample positions) over three whole 400 x 400 frames.  TEST INFRASTRUCTURE.
    python tools/rebalance_sweep.py DIR_WITH_teacher_{coarse,fine}.npz [...]      (through gpurun; ~1 minute per teacher)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch  # noqa: E402
import _pkg  # noqa: E402

_pkg.load()
import train_like as TL  # noqa: E402
from oracle import r2l_oracle as O  # noqa: E402
from efficient_nerf_amd import NeRFEngine, PRECISIONS  # noqa: E402
from efficient_nerf_amd.teacher import rebalanced_state, get_rays  # noqa: E402
from efficient_nerf_amd._lib import R2LError  # noqa: E402

H = 400
focal = O.focal_from_angle(H)
poses = [O.pose_spherical(30., -30., 4.), O.pose_spherical(150., -85., 4.), O.pose_spherical(-100., -5., 4.)]
GRID = [(2, t) for t in (2., 4., 6., 8., 11., 14., 20., 28., 32., 40., 56., 80.)] + [(0, 8.), (0, 14.), (0, 28.), (0, 56.), (1, 8.)]

with torch.no_grad():
    for d in sys.argv[1:]:
        sds = [TL.load_sd(os.path.join(d, f'teacher_{n}.npz')) for n in ('coarse', 'fine')]
        eng = NeRFEngine(H, H, focal, precision=PRECISIONS['fp16x3_asm']).load_state_dicts(*sds)
        refs, zs, rays = [], [], []
        for p in poses:
            ro, rd = (t.reshape(-1, 3) for t in get_rays(H, H, focal, p[:3, :4], device=eng.device))
            r = eng.render_rays(ro, rd, extras=True)
            refs.append(r['rgb_map'].clone())
            idx = torch.arange(0, H * H, 39, device=ro.device)          # the probes auto calibrates on: strided rays of each frame
            rays.append((ro[idx].contiguous(), rd[idx].contiguous()))
            zs.append(r['z_vals'][idx].clone())
        mx = eng.fine_activation_maxima(torch.cat([r[0] for r in rays]), torch.cat([r[1] for r in rays]), torch.cat(zs))
        print(f'# {d}: activation maxima of the fine network on the probes: ' + ', '.join(f'{k} {float(v):.1f}' for k, v in mx.items()), flush=True)

        def measure(sd, label):
            try:
                eng._load(1, sd)
            except R2LError as e:
                print(f'  {label}: not packable ({e})', flush=True)
                return
            eng.set_precision(PRECISIONS['fp16_mix'])
            out = []
            for p, ref in zip(poses, refs):
                dlt = (eng.render(p)['rgb_map'] - ref).abs().max(-1)[0]
                out.append((float(dlt.max()), int((dlt > 5e-5).sum())))
            eng.set_precision(PRECISIONS['fp16x3_asm'])
            if label:
                print(f'  {label}: rgb L_inf against three passes ' + ', '.join(f'{a:.2e} ({n} rays > 5e-5)' for a, n in out) + f' | worst {max(a for a, _ in out):.2e}', flush=True)
            return max(a for a, _ in out), sum(n for _, n in out)
        measure(sds[1], 'as loaded (no rebalancing)')
        for keep, target in GRID:
            sd, sh = rebalanced_state(sds[1], mx, keep=keep, target=target)
            measure(sd, f'keep {keep}, target {target:>4.1f}, shifts ' + ' '.join(f'{v:+d}' for v in sh.values()))
        # exploratory: coordinate descent on the ten shifts from target 32, objective = the worst ray of the three whole frames
        import math
        names = [f'h{i}' for i in range(8)] + ['feature', 'views']

        def with_shifts(sh):
            fake = {k: 8.0 * 2.0 ** sh[k] for k in names}       # rebalanced_state(target 8): s = ceil(log2(max / 8)) = sh
            return rebalanced_state(sds[1], fake, keep=0, target=8.0)[0]
        sh = {k: (0 if not float(mx[k]) > 0 else int(math.ceil(math.log2(float(mx[k]) / 32.)))) for k in names}
        best = measure(with_shifts(sh), None)
        print(f'  coordinate descent from target 32 (keep 0): start {best[0]:.2e}', flush=True)
        for rnd in range(2):
            for k in names:
                for dlt in (-1, +1):
                    trial = dict(sh, **{k: sh[k] + dlt})
                    r = measure(with_shifts(trial), None)
                    if r is not None and r[0] < best[0] * 0.97:
                        sh, best = trial, r
            print(f'    round {rnd}: worst {best[0]:.2e} ({best[1]} rays > 5e-5), shifts ' + ' '.join(f'{sh[k]:+d}' for k in names), flush=True)
        eng.close()
