#!/usr/bin/env python
"""Trained-like teacher, fine pass in fp16_fp8 with the coarse pass in fp16x3: does an exact power-of-two reparametrisation of the
fine network (relu is positively homogeneous: hidden activations / s, the consumers' weights x s) that brings its activations inside the
layer chain's fixed bf6 range (|a| <= 14) bring the error inside the contract?  Whole 400 x 400 frames against fp16x3 / fp16x3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import _pkg; _pkg.load()
from efficient_nerf_amd import NeRFEngine, PRECISIONS
from oracle import r2l_oracle as O
d = os.path.join(ROOT, 'tests', 'golden', 'trained_like')
ld = lambda n: {k: torch.from_numpy(v) for k, v in np.load(os.path.join(d, n)).items()}


def rescale(sd, s_trunk, s_feat, s_view):
    """h_i / s_trunk (i = 0..7), feature / s_feat, view-layer output / s_view; all exact for powers of two"""
    sd = {k: v.clone() for k, v in sd.items()}
    sd['pts_linears.0.weight'] /= s_trunk
    for i in range(8):
        sd[f'pts_linears.{i}.bias'] /= s_trunk
    sd['pts_linears.5.weight'][:, :63] /= s_trunk
    sd['alpha_linear.weight'] *= s_trunk
    sd['feature_linear.weight'] *= s_trunk / s_feat
    sd['feature_linear.bias'] /= s_feat
    sd['views_linears.0.weight'][:, :256] *= s_feat / s_view
    sd['views_linears.0.weight'][:, 256:] /= s_view
    sd['views_linears.0.bias'] /= s_view
    sd['rgb_linear.weight'] *= s_view
    return sd


c, f = ld('teacher_coarse.npz'), ld('teacher_fine.npz')
H = 400
focal = O.focal_from_angle(H)
poses = [O.pose_spherical(30., -30., 4.), O.pose_spherical(150., -85., 4.), O.pose_spherical(-100., -5., 4.)]
ref_eng = NeRFEngine(H, H, focal, precision=PRECISIONS['fp16x3']).load_state_dicts(c, f)
refs = [{k: v.clone() for k, v in ref_eng.render(p).items()} for p in poses]
for st, sf, sv in ((1, 1, 1), (4, 4, 1), (4, 4, 4), (8, 8, 2), (16, 16, 4), (2, 4, 1)):
    eng = NeRFEngine(H, H, focal, precision=PRECISIONS['fp16x3']).load_state_dicts(c, rescale(f, float(st), float(sf), float(sv)))
    same = max((eng.render(p)['rgb_map'] - r['rgb_map']).abs().max().item() for p, r in zip(poses, refs))
    eng.set_precision_pair(PRECISIONS['fp16x3'], PRECISIONS['fp16_fp8'])
    line = f'trunk / {st}, feature / {sf}, view / {sv}: fp16x3 of the rescaled network vs the original {same:.1e};  coarse fp16x3 + fine fp16_fp8:'
    for p, r in zip(poses, refs):
        g = eng.render(p)
        dd = (g['rgb_map'] - r['rgb_map']).abs().max(-1)[0]
        line += f'  rgb {dd.max().item():.2e} (>1e-4: {(dd > 1e-4).sum().item()}, >3e-5: {(dd > 3e-5).sum().item()}) depth {(g["depth_map"] - r["depth_map"]).abs().max().item():.1e}'
    print(line, flush=True)
    eng.close()
