import os, sys
sys.path.insert(0, '/root/repo')
import torch
import _pkg; _pkg.load()
from efficient_nerf_amd import NeRFEngine, PREC_FP16X1, PREC_FP16X3, PREC_FP16_FP8
from efficient_nerf_amd.teacher import get_rays
from oracle import r2l_oracle as O
H = 400; focal = O.focal_from_angle(H)
pose = O.novel_poses(200)[10][:3, :4]
for dens in (1.0, 10.0, 50.0, 200.0):
    sds = []
    for s in (1, 2):
        sd = O.make_teacher_state(s)
        sd['alpha_linear.weight'] = sd['alpha_linear.weight'] * dens
        sd['alpha_linear.bias'] = sd['alpha_linear.bias'] * dens
        sds.append(sd)
    eng = NeRFEngine(H, H, focal, white_bkgd=True, precision=PREC_FP16X3).load_state_dicts(*sds)
    ro, rd = get_rays(H, H, focal, pose, device=eng.device)
    name, diff = eng.choose_precision(ro.reshape(-1, 3), rd.reshape(-1, 3))
    probe = dict(eng.auto_diffs)
    outs = {}
    for n, p in (('fp16x3', PREC_FP16X3), ('fp16_fp8', PREC_FP16_FP8), ('fp16x1', PREC_FP16X1)):
        eng.set_precision(p); outs[n] = {k: v.clone() for k, v in eng.render(pose).items()}
    full = {n: (outs[n]['rgb_map'] - outs['fp16x3']['rgb_map']).abs().max().item() for n in ('fp16_fp8', 'fp16x1')}
    acc = outs['fp16x3']['acc_map']
    print(f'density x {dens:g}: auto -> {name}; probe diffs {probe}; whole frame rgb L_inf vs fp16x3: {full}; acc range {acc.min().item():.3f}..{acc.max().item():.3f}', flush=True)
    eng.close()
