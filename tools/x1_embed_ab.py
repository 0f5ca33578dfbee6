#!/usr/bin/env python
"""A/B of the fp16x1 teacher chain (four column tiles): embedding as HIP code between two tile blocks (round 4, nerf_chain_kernel<true, 4>)
against embedding, ray loads and raw stores inside the generated stream (round 5, nerf_chain_emb_kernel); same process, 400x400 frames,
alternating; every output and extra bitwise equal."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import _pkg; _pkg.load()
from efficient_nerf_amd import NeRFEngine, PREC_FP16X1
from efficient_nerf_amd._lib import lib, check
from oracle import r2l_oracle as O
H = int(sys.argv[1]) if len(sys.argv) > 1 else 400
eng = NeRFEngine(H, H, O.focal_from_angle(H), white_bkgd=True, precision=PREC_FP16X1).load_state_dicts(O.make_teacher_state(1), O.make_teacher_state(2))
pose = O.novel_poses(1)[0][:3, :4]
outs = {}
for rnd in range(3):
    for on in (0, 1):
        check(lib().nerf_debug_set_x1_stream_embed(eng._ctx, on))
        outs[on] = {k: v.clone() for k, v in eng.render(pose, extras=True).items()}
        torch.cuda.synchronize()
        eng.timing(True); eng.kernel_time_ms(reset=True)
        t0 = time.time()
        for _ in range(10):
            eng.render(pose)
        torch.cuda.synchronize()
        ms, n = eng.kernel_time_ms(reset=True); eng.timing(False)
        print(f'round {rnd}: embedding {"in the stream" if on else "as HIP code"}: {(time.time() - t0) * 100:.3f} ms per {H}x{H} frame, MLP kernels {ms / 10:.3f} ms', flush=True)
print('every output and extra bitwise equal:', all(torch.equal(outs[0][k], outs[1][k]) for k in outs[0]))
bad = [k for k in outs[0] if not torch.equal(outs[0][k], outs[1][k])]
for k in bad:
    print(k, (outs[0][k] - outs[1][k]).abs().max().item())
