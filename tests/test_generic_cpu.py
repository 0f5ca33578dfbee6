"""CPU side of the generic fp32 layer path (efficient-nerf_amd/generic.py): the oracle's restatement of ANY NeRF_v3_2 the
reference's constructor builds against the golden vectors generated from the reference's own classes
(tests/golden/make_golden_generic.py), and the host logic that maps the command-line flags to Linear layers and state_dict
keys (no compute call: no GPU here)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import r2l_oracle as O

T = torch.from_numpy


@pytest.fixture(scope='module')
def g(golden_dir):
    return np.load(os.path.join(golden_dir, 'r2l_generic.npz'))


def cases(g):
    return json.loads(str(g['cases']))


def embedded(g, cs):
    H = int(g['H'])
    ns, L = cs.get('n_sample', 16), cs.get('L', 10)
    c2w = T(g['c2w'])
    pts = O.sample_test(O.camera_dirs(H, H, float(g['focal'])), O.sampler_z_vals(ns, 2., 6.), c2w[:3, :4])[T(g['idx'])]
    return O.positional_embed(pts, L)


def test_oracle_reproduces_the_reference_on_every_variant(g):
    for cs in cases(g):
        emb = embedded(g, cs)
        np.testing.assert_allclose(emb[:8].numpy(), g[cs['name'] + '_emb8'], rtol=0, atol=2e-7)     # sin / cos: host libm
        sd = O.make_v3_2_state(int(g[cs['name'] + '_seed']), cs['netdepth'], cs['netwidth'], emb.shape[1], cs.get('layerwise_netwidths', ''),
                               cs.get('act', 'relu'), cs['trial'])
        rgb = O.v3_2_forward(sd, emb, cs['netdepth'], cs.get('act', 'relu'), cs.get('use_residual', True), cs['trial'])
        assert (rgb - T(g[cs['name'] + '_rgb'])).abs().max().item() <= 2e-6, cs['name']


def test_plan_names_the_reference_state_dict(g, pkg):
    """generic.v3_2_plan must ask for exactly the keys the reference's constructor creates, with their shapes, in order."""
    from efficient_nerf_amd.generic import v3_2_plan
    for cs in cases(g):
        input_dim = 3 * cs.get('n_sample', 16) * (2 * cs.get('L', 10) + 1)
        sd = O.make_v3_2_state(1, cs['netdepth'], cs['netwidth'], input_dim, cs.get('layerwise_netwidths', ''), cs.get('act', 'relu'), cs['trial'])
        plan = v3_2_plan(cs['netdepth'], cs['netwidth'], input_dim, 3, cs.get('layerwise_netwidths', ''), cs.get('act', 'relu'),
                         cs.get('use_residual', True), cs['trial'])
        names = [f"{p['key']}.{k}" for p in plan for k in ('weight', 'bias')]
        assert names == list(sd), cs['name']
        for p in plan:
            assert tuple(sd[p['key'] + '.weight'].shape) == (p['out_dim'], p['in_dim']), (cs['name'], p['key'])


def test_plan_refuses_what_the_reference_cannot_run(pkg):
    from efficient_nerf_amd import R2LError
    from efficient_nerf_amd.generic import v3_2_plan
    with pytest.raises(R2LError):      # widths that do not chain: head 64 -> ResMLP(128)
        v3_2_plan(8, 128, 1008, 3, '64,64,64,64,64,64,64', 'relu', True, dict(body_arch='resmlp'))
    with pytest.raises(R2LError):      # nn.Sequential of None
        v3_2_plan(8, 128, 1008, 3, '', 'none', True, dict(body_arch='mlp'))
    with pytest.raises(R2LError):
        v3_2_plan(8, 128, 1008, 3, '', 'gelu', True, None)
    with pytest.raises(R2LError):      # too few layer widths for the depth
        v3_2_plan(8, 128, 1008, 3, '64,64', 'relu', True, None)
    with pytest.raises(R2LError):      # global skip over different widths
        v3_2_plan(5, 64, 1008, 3, '64,32,48,48', 'relu', True, dict(body_arch='mlp'))
