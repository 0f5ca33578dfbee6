#!/usr/bin/env python
"""A/B of the fused coarse scan (raw2outputs + sample_pdf + merge in one launch) against the three stand-alone launches on
400x400 teacher frames (same context, nerf_debug_set_split_scans)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import _pkg; _pkg.load()
from efficient_nerf_amd import NeRFEngine, PREC_FP16_FP8
from efficient_nerf_amd._lib import lib, check
from oracle import r2l_oracle as O
H = 400
eng = NeRFEngine(H, H, O.focal_from_angle(H), white_bkgd=True, precision=PREC_FP16_FP8).load_state_dicts(O.make_teacher_state(1), O.make_teacher_state(2))
pose = O.novel_poses(1)[0][:3, :4]
for rnd in range(3):
    for split in (1, 0):
        check(lib().nerf_debug_set_split_scans(eng._ctx, split))
        eng.render(pose); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(10):
            eng.render(pose)
        torch.cuda.synchronize()
        print(f"round {rnd} {'three launches' if split else 'one fused launch'}: {(time.time() - t0) * 100:.3f} ms per 400x400 frame", flush=True)
