#!/usr/bin/env python
"""R2L_PREC_FP16_SPLIT on the trained-like student at 800 x 800: per split (head + blocks [0, split) in three passes, blocks
[split, 43) with bf6 terms -- fp16_split -- or e4m3 terms -- fp16_split8) the frame time and the L_inf / rms against three passes everywhere (fp16x3_asm) and against the CPU oracle
on every 8th row, then what `--precision auto` picks.  Beside it the whole-network modes (bf6-term head launch)."""
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
poses = [test[i][:3, :4] for i in (0, 67, 133)]
eng = R2LEngine(H, H, focal, 2., 6., n_block=43, use_residual=True, precision=PRECISIONS['fp16x3_asm']).load_state_dict(sd)
ref = [eng.render(p).clone() for p in poses]
want = [O.r2l_render(sd, H, H, focal, p, rows=(0, H, 8), chunk=16384) for p in poses]


def timed(n=10):
    eng.render(poses[0]); torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): eng.render(test[2 + i][:3, :4])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


print(f'fp16x3_asm: {timed():.2f} ms per frame')
def row(name):
    line = f'{name}: {timed():6.2f} ms per frame |'
    for p, r, w in zip(poses, ref, want):
        g = eng.render(p)
        d = (g - r).abs().max(-1)[0]
        line += (f' vs x3_asm {d.max().item():.2e} rms {(g - r).pow(2).mean().sqrt().item():.2e} (> 1e-4: {(d > 1e-4).sum().item()}, > 5e-5: {(d > 5e-5).sum().item()})'
                 f' vs oracle {(g.cpu().view(H, H, 3)[::8].reshape(-1, 3) - w).abs().max().item():.2e} |')
    print(line, flush=True)


eng.set_precision(PRECISIONS['fp16_fp8']); eng.calibrate_on(c2w=poses[0])
row('fp16_fp8 ')
for mode, splits in (('fp16_split', (0, 2, 5, 8, 12, 16, 20, 24, 28, 32, 38, 43)), ('fp16_split8', (0, 1, 2, 5, 8, 12, 16, 24, 32))):
    eng.set_precision(PRECISIONS[mode])
    print(mode + ': head + blocks [0, split) in three passes, blocks [split, 43) with ' + ('e4m3' if mode.endswith('8') else 'bf6') + ' terms')
    for sp in splits:
        eng.set_split_block(sp)
        row(f'split {sp:2d} ')
eng.close()
eng = R2LEngine(H, H, focal, 2., 6., n_block=43, use_residual=True).load_state_dict(sd)
t0 = time.perf_counter(); rung, top = eng.choose_precision(c2w=poses[0]); dt = time.perf_counter() - t0
print(f'auto: {rung} split {eng.split_block} (measured {eng.auto_split}; costs in bf6 blocks: ' + ', '.join(f'{m} {eng.split_cost(PRECISIONS[m], min(k for k, v in t.items() if v <= eng.AUTO_SPLIT_MAX_DIFF)):.1f}' for m, t in eng.auto_split.items() if any(v <= eng.AUTO_SPLIT_MAX_DIFF for v in t.values())) + f') in {dt:.2f} s; {timed():.2f} ms per frame = {H * H / timed() * 1e3:.3e} rays/s')
