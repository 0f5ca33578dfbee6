#!/usr/bin/env python
"""Why do the teacher's fast modes miss by 0.1-0.3 on the trained-like teacher (gpurun_out/trained_like or tests/golden/trained_like)?
raw of both networks (run_network on the fp16x3 render's own sample positions), then the composited maps, per mode against fp16x3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import _pkg; _pkg.load()
from efficient_nerf_amd import NeRFEngine, PRECISIONS
from oracle import r2l_oracle as O
d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'tests', 'golden', 'trained_like')
ld = lambda n: {k: torch.from_numpy(v) for k, v in np.load(os.path.join(d, n)).items()}
sds = (ld('teacher_coarse.npz'), ld('teacher_fine.npz'))
H = 400
focal = O.focal_from_angle(H)
eng = NeRFEngine(H, H, focal, precision=PRECISIONS['fp16x3']).load_state_dicts(*sds)
pose = O.pose_spherical(30., -30., 4.)
ref = {k: v.clone() for k, v in eng.render(pose, extras=True).items()}
from efficient_nerf_amd import get_rays
ro, rd = (t.reshape(-1, 3) for t in get_rays(H, H, focal, pose[:3, :4], device='cuda'))
zc = eng.z_coarse.cuda()
raw0 = eng.run_network(0, ro, rd, zc).clone()
raw1 = eng.run_network(1, ro, rd, ref['z_vals']).clone()
print('fp16x3: raw0 range', raw0.min().item(), raw0.max().item(), 'raw1', raw1.min().item(), raw1.max().item())
for pn in ('fp16x1', 'fp16_fp8'):
    eng.set_precision(PRECISIONS[pn])
    r0 = eng.run_network(0, ro, rd, zc)
    r1 = eng.run_network(1, ro, rd, ref['z_vals'])
    e0, e1 = (r0 - raw0).abs(), (r1 - raw1).abs()
    print(f'{pn}: raw coarse max abs err rgb {e0[..., :3].max().item():.3e} sigma {e0[..., 3].max().item():.3e} | fine rgb {e1[..., :3].max().item():.3e} sigma {e1[..., 3].max().item():.3e}'
          f' | sigma rel err at sigma>1: {(e1[..., 3] / raw1[..., 3].abs().clamp(min=1.)).max().item():.3e}')
    got = eng.render(pose, extras=True)
    for k in ('rgb_map', 'acc_map', 'depth_map', 'rgb0', 'acc0', 'z_samples'):
        dd = (got[k] - ref[k]).abs()
        dd = dd.reshape(dd.shape[0], -1).max(-1)[0]
        i = int(dd.argmax())
        print(f'   {k}: max {dd.max().item():.3e} at ray {i} (row {i // H}, col {i % H}); rays > 1e-3: {(dd > 1e-3).sum().item()}, > 1e-4: {(dd > 1e-4).sum().item()}')
    i = int((got['rgb_map'] - ref['rgb_map']).abs().max(-1)[0].argmax())
    print('   worst ray: acc', ref['acc_map'][i].item(), got['acc_map'][i].item(), 'acc0', ref['acc0'][i].item(), got['acc0'][i].item())
    w = ref['raw'][i, :, 3]
    print('   its fine sigma (fp16x3) top:', [round(float(x), 1) for x in w.topk(6)[0]], ' z_samples max shift', (got['z_samples'][i] - ref['z_samples'][i]).abs().max().item())
