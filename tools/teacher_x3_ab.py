#!/usr/bin/env python
"""fp16x3 of the teacher: compiler-scheduled nerf_mlp_kernel<2> against the generated three-pass chain (fp16x3_asm,
nerf_chain_kernel<false, 2, true>), same process, 400x400 frames; synthetic and trained-like weights; every output against fp16x3 and
rgb against the CPU oracle on 4,000 rays."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import _pkg; _pkg.load()
from efficient_nerf_amd import NeRFEngine, PRECISIONS
from oracle import r2l_oracle as O
H = 400
focal = O.focal_from_angle(H)
d = os.path.join(ROOT, 'tests', 'golden', 'trained_like')
ld = lambda n: {k: torch.from_numpy(v) for k, v in np.load(os.path.join(d, n)).items()}
torch.set_num_threads(16)
for label, sds in (('synthetic', (O.make_teacher_state(1), O.make_teacher_state(2))), ('trained-like', (ld('teacher_coarse.npz'), ld('teacher_fine.npz')))):
    eng = NeRFEngine(H, H, focal, precision=PRECISIONS['fp16x3']).load_state_dicts(*sds)
    poses = [O.pose_spherical(30., -30., 4.), O.pose_spherical(150., -85., 4.), O.pose_spherical(-100., -5., 4.)]
    outs = {}
    for name in ('fp16x3', 'fp16x3_asm'):
        eng.set_precision(PRECISIONS[name])
        outs[name] = [{k: v.clone() for k, v in eng.render(p, extras=True).items()} for p in poses]
        for rnd in range(2):
            eng.timing(True); eng.kernel_time_ms(reset=True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for p in poses:
                eng.render(p)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
            ms, n = eng.kernel_time_ms(reset=True); eng.timing(False)
            print(f'{label} {name}: {dt * 1e3:.2f} ms per frame, MLP kernels {ms / 3:.2f} ms, {H * H / dt:.3e} rays/s, {2 * 593408 * 256 * H * H / dt / 2.5e15:.3f} of the fp16 peak', flush=True)
    for i in range(3):
        line = f'{label} pose {i}: fp16x3_asm - fp16x3:'
        for k in ('raw', 'rgb_map', 'acc_map', 'depth_map', 'z_samples'):
            dd = (outs['fp16x3_asm'][i][k] - outs['fp16x3'][i][k]).abs()
            line += f' {k} {dd.max().item():.2e}'
        dr = (outs['fp16x3_asm'][i]['rgb_map'] - outs['fp16x3'][i]['rgb_map']).abs().max(-1)[0]
        print(line + f' | rays with rgb > 1e-5: {(dr > 1e-5).sum().item()}, > 1e-4: {(dr > 1e-4).sum().item()}', flush=True)
    idx = torch.arange(0, H * H, 40)
    ro, rd = O.get_rays(H, H, focal, poses[0][:3, :4])
    want = O.render_rays(sds[0], sds[1], ro.reshape(-1, 3)[idx].float(), rd.reshape(-1, 3)[idx].float(), white_bkgd=True)['rgb_map']
    for name in ('fp16x3', 'fp16x3_asm'):
        print(f'{label} {name} vs CPU oracle on {len(idx)} rays: {(outs[name][0]["rgb_map"].cpu()[idx] - want).abs().max().item():.2e}', flush=True)
    eng.close()
