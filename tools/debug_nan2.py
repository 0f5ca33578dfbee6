import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, numpy as np
import _pkg; _pkg.load()
from efficient_nerf_amd import R2LEngine, PREC_FP16X1, PREC_FP16X3
from oracle import r2l_oracle as O
H = 8
poses = {'rand': O.rand_poses(2, seed=11)[1], 'fixed': O.pose_spherical(-37.8, -30., 4.)}
for nb in (0, 1, 43):
    for seed in (0, 5):
        sd = O.make_r2l_state(seed=seed, netdepth=2 + 2 * nb)
        for res in (True, False):
            eng = R2LEngine(H, H, O.focal_from_angle(H), n_block=nb, use_residual=res).load_state_dict(sd)
            for pn, c2w in poses.items():
                rgb = eng.render(c2w).cpu()
                dirs = O.camera_dirs(H, H, O.focal_from_angle(H))
                pts = O.sample_test(dirs, O.sampler_z_vals(16, 2., 6.), c2w[:3, :4])
                ref = O.r2l_forward(sd, O.positional_embed(pts, 10), use_residual=res)
                err = (rgb - ref).abs()
                print(f'nb={nb} seed={seed} res={res} pose={pn}: nan={int(torch.isnan(rgb).sum())} maxerr={err[~torch.isnan(err)].max().item() if (~torch.isnan(err)).any() else -1:.3e} rgb0={rgb[0].tolist()} ref0={ref[0].tolist()}', flush=True)
            eng.close()
