#!/usr/bin/env python
"""CPU study (no GPU): which layers of the trained-like teacher need three fp16 passes?  The density path (trunk -> alpha) is sharp (sigma up
to 200) and steers the fine samples; the view branch (feature_linear, views_linears.0, rgb_linear: 17 % of the MACs) only colours.
Emulated single fp16 pass = both operands of a layer rounded to fp16, products and sums in fp32.  Prints the L_inf of the composited maps
against the all-fp32 oracle on rays spread over a 400 x 400 frame, for: view branch in one pass; trunk in one pass; everything in one pass."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import torch.nn.functional as F
from oracle import r2l_oracle as O

d = os.path.join(ROOT, 'tests', 'golden', 'trained_like')
ld = lambda n: {k: torch.from_numpy(v) for k, v in np.load(os.path.join(d, n)).items()}
sds = (ld('teacher_coarse.npz'), ld('teacher_fine.npz'))
H = 400
focal = O.focal_from_angle(H)
q = lambda t: t.half().float()
exact = O.teacher_forward


def forward(one_pass):
    def f(sd, x, input_ch=63, skips=(4,), dtype=torch.float32):
        lin = lambda name, h: (F.linear(q(h), q(sd[name + '.weight']), sd[name + '.bias']) if name.split('.')[0] in one_pass
                               else F.linear(h, sd[name + '.weight'], sd[name + '.bias']))
        input_pts, input_views = x[..., :input_ch], x[..., input_ch:]
        h = input_pts
        for i in range(8):
            h = F.relu(lin(f'pts_linears.{i}', h))
            if i in skips:
                h = torch.cat([input_pts, h], -1)
        alpha = lin('alpha_linear', h)
        feature = lin('feature_linear', h)
        h = F.relu(lin('views_linears.0', torch.cat([feature, input_views], -1)))
        return torch.cat([lin('rgb_linear', h), alpha], -1)
    return f


n = int(os.environ.get('N_RAYS', 8000))
for pose in (O.pose_spherical(30., -30., 4.), O.pose_spherical(150., -85., 4.), O.pose_spherical(-100., -5., 4.)):
    ro, rd = O.get_rays(H, H, focal, pose[:3, :4])
    idx = torch.arange(0, H * H, H * H // n)[:n]
    ro, rd = ro.reshape(-1, 3)[idx].float(), rd.reshape(-1, 3)[idx].float()
    O.teacher_forward = exact
    ref = O.render_rays(sds[0], sds[1], ro, rd, white_bkgd=True)
    for tag, sel in (('view branch (feature, views, rgb) in one pass', ('feature_linear', 'views_linears', 'rgb_linear')),
                     ('views + rgb in one pass', ('views_linears', 'rgb_linear')),
                     ('trunk + alpha in one pass', ('pts_linears', 'alpha_linear')),
                     ('everything in one pass', ('pts_linears', 'alpha_linear', 'feature_linear', 'views_linears', 'rgb_linear'))):
        O.teacher_forward = forward(sel)
        got = O.render_rays(sds[0], sds[1], ro, rd, white_bkgd=True)
        print(f'{tag}: ' + ', '.join(f'{k} {(got[k] - ref[k]).abs().max().item():.2e}' for k in ('rgb_map', 'acc_map', 'depth_map', 'rgb0')), flush=True)
    print()
O.teacher_forward = exact
