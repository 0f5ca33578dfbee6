#!/usr/bin/env python
"""The trained-like student (tests/golden/trained_like/student_w256d88.npz, max|a| 126) forced through every R2L mode: measured L_inf
against fp16x3_asm over whole 800 x 800 frames and against the CPU oracle on every 8th row -- are the ladder's limits (fp16_fp8 up to
max|a| 8, fp16_e4m3 up to 10: derived from i.i.d. weight families) what a trained network needs?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import _pkg; _pkg.load()
from efficient_nerf_amd import PRECISIONS, R2LEngine
from oracle import r2l_oracle as O
z = np.load(os.path.join(ROOT, 'tests', 'golden', 'trained_like', 'student_w256d88.npz'))
sd = {k: torch.from_numpy(z[k]) for k in z.files}
H = 800
focal = O.focal_from_angle(H)
test = O.novel_poses(200)
torch.set_num_threads(16)
eng = R2LEngine(H, H, focal, 2., 6., n_block=43, use_residual=True, precision=PRECISIONS['fp16x3_asm']).load_state_dict(sd)
poses = [test[i][:3, :4] for i in (0, 67, 133)]
ref = [eng.render(p).clone() for p in poses]
want = [O.r2l_render(sd, H, H, focal, p, rows=(0, H, 8), chunk=16384) for p in poses]
print('fp16x3_asm vs CPU oracle:', ['%.2e' % (r.cpu().view(H, H, 3)[::8].reshape(-1, 3) - w).abs().max().item() for r, w in zip(ref, want)])
for name in ('fp16_e4m3', 'fp16_fp8', 'fp16x1'):
    eng.set_precision(PRECISIONS[name])
    if name != 'fp16x1':
        eng.calibrate_on(c2w=poses[0])
        print(name, 'exponents', min(eng.act_exponents()), '..', max(eng.act_exponents()))
    line = name + ':'
    for p, r, w in zip(poses, ref, want):
        g = eng.render(p)
        d = (g - r).abs().max(-1)[0]
        line += f'  vs fp16x3_asm {d.max().item():.2e} (rays > 1e-4: {(d > 1e-4).sum().item()}, > 5e-5: {(d > 5e-5).sum().item()}), vs oracle {(g.cpu().view(H, H, 3)[::8].reshape(-1, 3) - w).abs().max().item():.2e}'
    if name != 'fp16x1':
        st = eng.range_status()
        line += f' | fills h0 {st["h0_fill"]:.2f} worst {st["worst_fill"]:.2f} saturated {st["saturated"]}'
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(5): eng.render(poses[i % 3])
    torch.cuda.synchronize()
    print(line, f'| {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per frame', flush=True)
