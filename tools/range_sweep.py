"""How the error of the low-precision correction terms grows with the magnitude of the residual stream: W256D88 networks
with every body weight scaled by a gain, 200x200 frames, exponents measured on every ray of the first pose; prints the
largest measured activation exponent (R2LEngine.act_exponents: |a| <= 2^E) and L_inf of fp16_fp8 (bf6 terms),
fp16_e4m3 (e4m3 terms) and fp16x3_asm (three fp16 passes on the generated kernels) against the compiler-scheduled fp16x3
(itself 6e-7 from the fp32 oracle), and what `--precision auto` picks.
Basis of R2LEngine.AUTO_MAX_EXP / AUTO_MAX_EXP_E4M3.  Run through gpurun: python tools/range_sweep.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import _pkg
_pkg.load()
from efficient_nerf_amd import R2LEngine, PREC_FP16X3, PREC_FP16_FP8, PREC_FP16_E4M3, PREC_FP16X3_ASM
from oracle import r2l_oracle as O

H = 200
focal = O.focal_from_angle(H)
for seed in (0, 1, 2):
    for gain in (1.0, 1.05, 1.1, 1.15, 1.2, 1.25, 1.3, 1.4, 1.5, 1.6):
        sd = O.make_r2l_state(seed=seed)
        for k in sd:
            if 'body' in k and k.endswith('weight'):
                sd[k] = sd[k] * gain
        poses = [O.pose_spherical(th, -30., 4.) for th in (0., 120., 240.)]
        e3 = R2LEngine(H, H, focal, precision=PREC_FP16X3).load_state_dict(sd)
        ref = [e3.render(c).cpu() for c in poses]
        e3.close()
        worst, top = {}, None
        for name, prec in (('fp16_fp8', PREC_FP16_FP8), ('fp16_e4m3', PREC_FP16_E4M3)):
            e8 = R2LEngine(H, H, focal, precision=prec).load_state_dict(sd)
            ex = e8.calibrate_on(c2w=poses[0])
            top = (max(ex[0::2]), max(ex[1::2]), e8.stream_max)
            worst[name] = max((e8.render(c).cpu() - r).abs().max().item() for c, r in zip(poses, ref))
            e8.close()
        ex = R2LEngine(H, H, focal, precision=PREC_FP16X3_ASM).load_state_dict(sd)
        worst['fp16x3_asm'] = max((ex.render(c).cpu() - r).abs().max().item() for c, r in zip(poses, ref))
        ex.close()
        ea = R2LEngine(H, H, focal).load_state_dict(sd)
        pick = ea.choose_precision(c2w=poses[0])[0]
        err = max((ea.render(c).cpu() - r).abs().max().item() for c, r in zip(poses, ref))
        ea.close()
        print('seed %d gain %.2f: max exponent x %2d  h %2d  max|a| %5.1f   L_inf vs fp16x3: fp16_fp8 %.2e  fp16_e4m3 %.2e  fp16x3_asm %.2e   auto -> %-10s %.2e'
              % (seed, gain, top[0], top[1], top[2], worst['fp16_fp8'], worst['fp16_e4m3'], worst['fp16x3_asm'], pick, err), flush=True)
