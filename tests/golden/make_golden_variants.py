"""Golden vectors for the NeRF_v3_2 variants the reference constructor accepts beyond the README's network
(model/nerf_raybased.py:468-476, 483-537): other activations (act / trial.inact / trial.outact in relu, lrelu, none),
trial.res_scale, and trial.body_arch = mlp.  Runs the REFERENCE's own class on CPU (build container only):

    python tests/golden/make_golden_variants.py

Every case: a seeded model of the reference (its own constructor order and nn.Linear init), 192 embedded rays of a fixed
pose -> the reference's rgb; while generating, the oracle (oracle/r2l_oracle.py) must reproduce the state_dict bit for bit
and the output within 2e-6.  Only tests/golden/r2l_variants.npz travels."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('R2L_REFERENCE', '/root/reference')
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import model.nerf_raybased as RM  # noqa: E402  (reference)
from oracle import r2l_oracle as O  # noqa: E402

torch.set_grad_enabled(False)
torch.autograd.set_detect_anomaly(False)

CASES = [  # name, netdepth, body_arch, act, inact, outact, res_scale, seed
    ('lrelu_all', 12, 'resmlp', 'lrelu', 'lrelu', 'none', 1.0, 3),
    ('outact_relu_half', 12, 'resmlp', 'relu', 'relu', 'relu', 0.5, 4),
    ('mixed', 10, 'resmlp', 'lrelu', 'none', 'lrelu', 0.3, 5),
    ('mlp_relu', 8, 'mlp', 'relu', 'relu', 'none', 1.0, 6),
    ('mlp_lrelu', 10, 'mlp', 'lrelu', 'relu', 'none', 1.0, 7),
]


def main():
    H = 24
    focal = O.focal_from_angle(H)
    c2w = O.pose_spherical(25., -40., 4.)
    pts = O.sample_test(O.camera_dirs(H, H, focal), O.sampler_z_vals(16, 2., 6.), c2w[:3, :4])
    idx = torch.arange(0, H * H, 3)[:192]
    emb = RM.PositionalEmbedder(L=10)(pts[idx])
    assert torch.equal(emb, O.positional_embed(pts[idx], 10))
    out = dict(c2w=c2w.numpy(), H=np.int32(H), focal=np.float64(focal), idx=idx.numpy())
    for name, D, arch, act, inact, outact, rs, seed in CASES:
        args = SimpleNamespace(netdepth=D, netwidth=256, layerwise_netwidths='', act=act, linear_tail=False, use_residual=True,
                               trial=SimpleNamespace(body_arch=arch, n_block=-1, n_learnable=2, res_scale=rs, inact=inact, outact=outact))
        torch.manual_seed(seed)
        model = RM.NeRF_v3_2(args, 1008, 3).eval()
        sd_ref = {k: v.clone() for k, v in model.state_dict().items()}
        ref = model(emb)
        if arch == 'mlp':
            sd = O.make_r2l_mlp_state(seed, netdepth=D)
            mine = O.r2l_forward_mlp(sd, emb, act=act)
        else:
            sd = O.make_r2l_state(seed, netdepth=D, inact=inact)
            mine = O.r2l_forward(sd, emb, res_scale=rs, act=act, inact=inact, outact=outact)
        assert list(sd) == list(sd_ref), (name, list(sd)[:6], list(sd_ref)[:6])
        for k in sd:
            assert torch.equal(sd[k], sd_ref[k]), (name, k)
        err = (mine - ref).abs().max().item()
        assert err <= 2e-6, (name, err)
        out[name + '_rgb'] = ref.numpy()
        out[name + '_cfg'] = np.array([name, str(D), arch, act, inact, outact, repr(rs), str(seed)])
        print(f'{name}: D={D} {arch} act={act} inact={inact} outact={outact} res_scale={rs}: oracle - reference {err:.1e}; rgb range '
              f'{ref.min().item():.3f} .. {ref.max().item():.3f}')
    np.savez_compressed(os.path.join(HERE, 'r2l_variants.npz'), **out)
    print('wrote', os.path.join(HERE, 'r2l_variants.npz'))


if __name__ == '__main__':
    main()
