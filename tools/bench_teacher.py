#!/usr/bin/env python
"""Development bench (GPU box): NeRF teacher coarse+fine render of one 400x400 frame
(BASELINE.json config 3): rays/s and algorithmic TFLOP/s (303,824,896 FLOP/ray)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import _pkg  # noqa: E402

_pkg.load()
from efficient_nerf_amd import NeRFEngine, PREC_FP16X1, PREC_FP16X3  # noqa: E402
from oracle import r2l_oracle as O  # noqa: E402

H = W = int(os.environ.get('T_H', 400))
REP = int(os.environ.get('T_REP', 3))
FLOP_PER_RAY = 2 * 593408 * 256
ONLY = os.environ.get('T_PREC')  # e.g. fp16_fp8: that mode only (profiling runs)
for prec, name in ((PREC_FP16X3, 'fp16x3'), (2, 'fp16_fp8'), (PREC_FP16X1, 'fp16x1'), (4, 'fp16x3_asm'), (7, 'fp16_mix')):
    if ONLY and name != ONLY:
        continue
    eng = NeRFEngine(H, W, O.focal_from_angle(W), precision=prec).load_state_dicts(O.make_teacher_state(1),
                                                                                  O.make_teacher_state(2))
    poses = O.novel_poses(4)
    eng.render(poses[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(REP):
        out = eng.render(poses[i % 4])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / REP
    passes = {'fp16x3': 3, 'fp16_fp8': 1.5, 'fp16x1': 1, 'fp16x3_asm': 3, 'fp16_mix': 0.25 * 3 + 0.75 * 1.96}[name]   # mix: coarse three passes, fine 2 of ~8.6 layers
    print(f'teacher {name} {H}x{W}: {dt*1e3:.1f} ms/frame, {H*W/dt:.3e} rays/s, algorithmic {FLOP_PER_RAY*H*W/dt/1e12:.0f} TFLOP/s '
          f'(executed {FLOP_PER_RAY*H*W*passes/dt/1e12:.0f})', flush=True)
    eng.close()
