#!/usr/bin/env python
"""Bitwise repeatability of the teacher's modes: the same 400x400 frame rendered N times per mode (fp16x1 = the generated chain
without correction terms and, since round 5, with its embedding in the stream; fp16_fp8 = with the bf6 terms; fp16x3_asm = in three passes), every output compared with the first render."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import _pkg; _pkg.load()
from efficient_nerf_amd import NeRFEngine, PRECISIONS
from oracle import r2l_oracle as O
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
H = 400
poses = O.novel_poses(4)
for name in ('fp16x1', 'fp16_fp8', 'fp16x3_asm'):
    eng = NeRFEngine(H, H, O.focal_from_angle(H), white_bkgd=True, precision=PRECISIONS[name]).load_state_dicts(O.make_teacher_state(1), O.make_teacher_state(2))
    first = [{k: v.clone() for k, v in eng.render(p[:3, :4], extras=True).items()} for p in poses]
    bad = 0
    for i in range(N):
        out = eng.render(poses[i % 4][:3, :4], extras=True)
        bad += sum(int(not torch.equal(out[k], first[i % 4][k])) for k in out)
    print(f'teacher {name}: {N} frames of {H}x{H} over 4 poses, every output and extra against the first render of its pose: {bad} differing tensors', flush=True)
    eng.close()

# round 6: the mixed rung and the two exits of nerf_set_skip_rgb0 on the trained-like teacher (the second exit is a workgroup-uniform branch on an
# LDS word the four waves OR into, followed by a drain and a re-prime of the weight ring: a race there would show as a differing frame)
from oracle import whole_frame as WF
sds = WF.load_teacher()
for name in ('fp16_mix', 'fp16x3_asm'):
    eng = NeRFEngine(H, H, O.focal_from_angle(H), white_bkgd=True, precision=PRECISIONS[name]).load_state_dicts(*sds)
    eng.set_skip_rgb0(True)
    first = [{k: v.clone() for k, v in eng.render(p[:3, :4], extras=True).items()} for p in poses]
    bad = 0
    for i in range(N):
        out = eng.render(poses[i % 4][:3, :4], extras=True)
        bad += sum(int(not torch.equal(out[k].view(torch.int32), first[i % 4][k].view(torch.int32))) for k in out)
    print(f'trained-like teacher {name} with nerf_set_skip_rgb0 (coarse pass without its view branch, fine pass with the second exit): {N} frames of {H}x{H} over 4 poses, '
          f'every output and extra against the first render of its pose: {bad} differing tensors', flush=True)
    eng.close()
