import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
import _pkg; _pkg.load()
from efficient_nerf_amd import PREC_FP16_FP8, R2LEngine
from oracle import r2l_oracle as O
for nb in (1, 2, 3, 5):
    sd = O.make_r2l_state(seed=1, netdepth=2 + 2 * nb)
    eng = R2LEngine(8, 8, O.focal_from_angle(8), n_block=nb, precision=PREC_FP16_FP8).load_state_dict(sd)
    for nt in (1, 3):
        x = (torch.rand(nt, 4, 32, 64, 4, generator=torch.Generator().manual_seed(1)) * 16).cuda()
        out = eng.debug_body(x).cpu()
        nan = torch.isnan(out)
        print('nb', nb, 'tiles', nt, 'nan', int(nan.sum()), 'of', out.numel(), 'exps', eng.act_exponents()[:6],
              'nan per wave', [int(nan[:, w].sum()) for w in range(4)], 'per group0..3', [int(nan[:, :, g].sum()) for g in range(4)], flush=True)
    rgb = eng.render(O.pose_spherical(10., -30., 4.)).cpu()
    print('   render nan', int(torch.isnan(rgb).sum()))
    eng.close()
