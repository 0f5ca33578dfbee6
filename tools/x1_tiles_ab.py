#!/usr/bin/env python
"""A/B of the fp16x1 teacher chain with three against two 16-point column tiles per wave (192- against 128-point workgroup tiles;
nerf_debug_set_x1_col_tiles), same process, 400x400 frames; and the two renders against each other (the arithmetic per point is
the same: bitwise equal)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import _pkg; _pkg.load()
from efficient_nerf_amd import NeRFEngine, PREC_FP16X1
from efficient_nerf_amd._lib import lib, check
from oracle import r2l_oracle as O
H = 400
eng = NeRFEngine(H, H, O.focal_from_angle(H), white_bkgd=True, precision=PREC_FP16X1).load_state_dicts(O.make_teacher_state(1), O.make_teacher_state(2))
pose = O.novel_poses(1)[0][:3, :4]
outs = {}
for rnd in range(3):
    for nc in (2, 3, 4):
        check(lib().nerf_debug_set_x1_col_tiles(eng._ctx, nc))
        outs[nc] = {k: v.clone() for k, v in eng.render(pose, extras=True).items()}
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(10):
            eng.render(pose)
        torch.cuda.synchronize()
        print(f'round {rnd}: {nc} column tiles per wave: {(time.time() - t0) * 100:.3f} ms per 400x400 frame', flush=True)
check(lib().nerf_debug_set_x1_col_tiles(eng._ctx, 3))
print('every output and extra bitwise equal between the tilings:', all(torch.equal(outs[2][k], outs[3][k]) and torch.equal(outs[2][k], outs[4][k]) for k in outs[2]))
