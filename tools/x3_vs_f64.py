"""Which of the two fp32-grade modes is closer to the truth on networks with a large residual stream?  fp16x3 (compiler-
scheduled) and fp16x3_asm (generated head + body) against a float64 evaluation of the network on 4,000 rays of a 200x200 frame,
with the fp32 CPU oracle beside them; body weights x gain (tools/range_sweep.py's networks).  Run through gpurun."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import _pkg
_pkg.load()
from efficient_nerf_amd import R2LEngine, PRECISIONS
from oracle import r2l_oracle as O



def forward64(sd, emb):
    """the W256D88 network (model/nerf_raybased.py:443-465, 539-544) in float64"""
    g = lambda k: sd[k].double()
    h0 = torch.relu(emb.double() @ g('head.0.weight').T + g('head.0.bias'))
    x = h0
    nb = len([k for k in sd if k.endswith('body.0.weight')])
    for i in range(nb):
        h = torch.relu(x @ g(f'body.{i}.body.0.weight').T + g(f'body.{i}.body.0.bias'))
        x = x + h @ g(f'body.{i}.body.2.weight').T + g(f'body.{i}.body.2.bias')
    return torch.sigmoid((x + h0) @ g('tail.0.weight').T + g('tail.0.bias'))


H = 200
focal = O.focal_from_angle(H)
c2w = O.pose_spherical(120., -30., 4.)
idx = torch.arange(0, H * H, 10)
for gain in (1.0, 1.2, 1.3, 1.4, 1.5, 1.6):
    sd = O.make_r2l_state(seed=0)
    for k in sd:
        if 'body' in k and k.endswith('weight'):
            sd[k] = sd[k] * gain
    pts = O.sample_test(O.camera_dirs(H, H, focal), O.sampler_z_vals(16, 2., 6.), torch.as_tensor(c2w)[:3, :4])[idx]
    emb32 = O.positional_embed(pts, 10)
    ref32 = O.r2l_forward(sd, emb32)
    ref64 = forward64(sd, emb32)        # the embedding as the reference computes it (fp32), the network in float64
    out = {}
    for name in ('fp16x3', 'fp16x3_asm'):
        e = R2LEngine(H, H, focal, precision=PRECISIONS[name]).load_state_dict(sd)
        out[name] = e.render(c2w).cpu()[idx].double()
        e.close()
    print('gain %.1f: L_inf against float64: fp32 CPU oracle %.2e   fp16x3 %.2e   fp16x3_asm %.2e' %
          (gain, (ref32.double() - ref64).abs().max().item(), (out['fp16x3'] - ref64).abs().max().item(),
           (out['fp16x3_asm'] - ref64).abs().max().item()), flush=True)
