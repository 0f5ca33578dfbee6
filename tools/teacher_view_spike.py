#!/usr/bin/env python
"""A teacher whose fp16x1 error depends on where the camera is: `rows` units of the view layer get weight -g on the raw d_z column
and bias -0.7 g, i.e. they are dead for |d_z| < 0.7 (horizontal cameras) and carry activations up to 0.3 g for cameras looking
down -- the construction behind tests/test_teacher_watch_gpu.py (a probe pose that passes, a pose that fails).  Prints the
whole-frame difference of fp16x1 / fp16_fp8 from fp16x3 for a horizontal and a top-down pose per gain."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import _pkg; _pkg.load()
from efficient_nerf_amd import NeRFEngine, PREC_FP16X1, PREC_FP16X3, PREC_FP16_FP8
from oracle import r2l_oracle as O


def spiked_teacher(seed, g, rows=64):
    sd = {k: v.clone() for k, v in O.make_teacher_state(seed).items()}
    sd['views_linears.0.weight'][:rows, 256 + 2] = -float(g)      # raw d_z column of the view embedding (input first, then sin / cos)
    sd['views_linears.0.bias'][:rows] -= 0.7 * float(g)
    return sd


if __name__ == '__main__':
    H = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    focal = O.focal_from_angle(H)
    poses = {'horizontal': O.pose_spherical(30., -5., 4.)[:3, :4], 'top-down': O.pose_spherical(150., -88., 4.)[:3, :4]}
    for g in (0, 16, 32, 64, 128, 256):
        eng = NeRFEngine(H, H, focal, precision=PREC_FP16X3).load_state_dicts(spiked_teacher(1, g), spiked_teacher(2, g))
        for name, pose in poses.items():
            eng.set_precision(PREC_FP16X3)
            ref = {k: v.clone() for k, v in eng.render(pose).items()}
            line = f'g={g:4d} {name:10s}'
            for pn, prec in (('fp16x1', PREC_FP16X1), ('fp16_fp8', PREC_FP16_FP8)):
                eng.set_precision(prec)
                got = eng.render(pose)
                line += f' | {pn}: ' + ' '.join(f'{k[:-4]} {(got[k] - ref[k]).abs().max().item():.1e}' for k in ('rgb_map', 'acc_map', 'depth_map'))
            print(line, flush=True)
        eng.close()
