import os, sys
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/_pkg.py') else os.getcwd())
import torch, numpy as np
import _pkg; _pkg.load()
from efficient_nerf_amd import R2LEngine, PREC_FP16X3, PREC_FP16_FP8, PREC_FP16X1
from oracle import r2l_oracle as O
H = 64
focal = O.focal_from_angle(H)
c2w = O.pose_spherical(30., -30., 4.)
for kind in ('uniform', 'laplace', 'sparse', 'outlier'):
    sd = O.make_r2l_state(seed=21, netdepth=88)
    g = torch.Generator().manual_seed(5)
    for k, w in sd.items():
        if not k.endswith('weight'): continue
        std = w.std()
        if kind == 'laplace':
            u = torch.rand(w.shape, generator=g) - 0.5
            sd[k] = (-torch.sign(u) * torch.log1p(-2 * u.abs()) * std / np.sqrt(2)).float()
        elif kind == 'sparse':
            m = (torch.rand(w.shape, generator=g) < 0.5).float()
            sd[k] = w * m * np.sqrt(2)
        elif kind == 'outlier':
            w2 = w.clone(); idx = torch.randint(0, w.numel(), (w.numel() // 2000,), generator=g)
            w2.view(-1)[idx] *= 12.0
            sd[k] = w2
    ref = O.r2l_render(sd, H, H, focal, c2w)
    ref64 = O.r2l_render(sd, H, H, focal, c2w, dtype=torch.float64)
    res = {}
    for name, prec in (('fp16x3', PREC_FP16X3), ('fp16_fp8', PREC_FP16_FP8), ('fp16x1', PREC_FP16X1)):
        eng = R2LEngine(H, H, focal, n_block=43, precision=prec).load_state_dict(sd)
        out = eng.render(c2w).cpu()
        res[name] = ((out - ref).abs().max().item(), (out.double() - ref64.double()).abs().max().item())
        eng.close()
    print(kind, 'fp32-oracle vs fp64', (ref.double() - ref64.double()).abs().max().item(), {k: (f'{a:.2e}', f'{b:.2e}') for k, (a, b) in res.items()}, 'rgb range', ref.min().item(), ref.max().item(), flush=True)
