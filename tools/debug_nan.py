import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, numpy as np
import _pkg; _pkg.load()
from efficient_nerf_amd import R2LEngine, PREC_FP16X1, PREC_FP16X3
from oracle import r2l_oracle as O
H = int(os.environ.get('DBG_H', 40)); nb = int(os.environ.get('DBG_NB', 1))
sd = O.make_r2l_state(seed=5, netdepth=2 + 2 * nb)
eng = R2LEngine(H, H, O.focal_from_angle(H), n_block=nb).load_state_dict(sd)
c2w = O.rand_poses(2, seed=11)[1]
for prec in (PREC_FP16X3, PREC_FP16X1):
    eng.set_precision(prec)
    for rep in range(3):
        rgb = eng.render(c2w).cpu()
        bad = torch.isnan(rgb).any(-1) | (rgb.abs() > 10).any(-1)
        idx = bad.nonzero().flatten().numpy()
        ref = O.r2l_render(sd, H, H, O.focal_from_angle(H), c2w)
        err = (rgb - ref).abs().max(-1)[0]
        wrong = (err > 1e-3) | bad
        w = wrong.nonzero().flatten().numpy()
        print(f'prec={prec} rep={rep} n={rgb.shape[0]} nan={len(idx)} wrong={len(w)}', flush=True)
        if len(w):
            tiles = np.unique(w // 128); waves = np.unique((w % 128) // 32); lanes = np.unique(w % 32)
            print('  tiles', tiles[:20], 'waves', waves, 'lanes', lanes[:40])
