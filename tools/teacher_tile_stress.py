#!/usr/bin/env python
"""Random shapes through the teacher's fp16x1 chain: ray counts 1 .. 6,000 (tile edges of 64 / 128 / 192 / 256 points included), random
sample counts, random weight seeds -- the four-column-tile render must equal the two- and three-tile renders bit for bit (the
arithmetic per point is the same: only the tiling differs) and stay within 1e-4 of fp16x3 on rgb."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import _pkg; _pkg.load()
from efficient_nerf_amd import NeRFEngine, PREC_FP16X1, PREC_FP16X3
from efficient_nerf_amd._lib import lib, check
from oracle import r2l_oracle as O
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(11)
bad = worst = 0
for it in range(N):
    S0 = int(rng.choice([3, 8, 16, 33, 64])); NI = int(rng.choice([1, 5, 32, 64, 128, 192])); NI = min(NI, 256 - S0)
    n = int(rng.choice([1, 2, 3, 4, 5, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 1000, 4099, int(rng.randint(1, 6000))]))
    seed = int(rng.randint(1, 1000))
    eng = NeRFEngine(8, 8, 10., N_samples=S0, N_importance=NI, white_bkgd=bool(it & 1), precision=PREC_FP16X3).load_state_dicts(O.make_teacher_state(seed), O.make_teacher_state(seed + 1))
    g = torch.Generator().manual_seed(it)
    ro = (torch.randn(n, 3, generator=g) * 0.3 + torch.tensor([0., 0., 4.])).cuda()
    rd = torch.nn.functional.normalize(torch.randn(n, 3, generator=g) * 0.2 + torch.tensor([0., 0., -1.]), dim=-1).cuda() * (0.7 + 0.6 * torch.rand(n, 1, generator=g).cuda())
    ref = eng.render_rays(ro, rd)['rgb_map'].clone()
    eng.set_precision(PREC_FP16X1)
    outs = {}
    for nc in (2, 3, 4):
        check(lib().nerf_debug_set_x1_col_tiles(eng._ctx, nc))
        outs[nc] = {k: v.clone() for k, v in eng.render_rays(ro, rd, extras=True).items()}
    check(lib().nerf_debug_set_x1_col_tiles(eng._ctx, 4))
    same = all(torch.equal(outs[2][k], outs[4][k]) and torch.equal(outs[3][k], outs[4][k]) for k in outs[2])
    d = (outs[4]['rgb_map'] - ref).abs().max().item()
    worst = max(worst, d)
    if not same or not d <= 1e-4:
        bad += 1
        print(f'case {it}: n={n} S0={S0} NI={NI} seed={seed}: tilings equal {same}, rgb vs fp16x3 {d:.2e}', flush=True)
    eng.close()
print(f'{N} random cases: {bad} bad; worst rgb difference from fp16x3 {worst:.2e}')
