"""CPU: decision logic of the teacher's `--precision auto` that needs no kernel (efficient-nerf_amd/teacher.py): the exact power-of-two
rebalancing of NeRF(D=8, W=256) (model/nerf_raybased.py:377-401) and the far-plane tie rule of the map comparisons (main.py:571-573:
the last sample's distance is 1e10, so its alpha is the SIGN of its raw density)."""
import torch

from oracle import r2l_oracle as O


def _engine():
    from efficient_nerf_amd.teacher import NeRFEngine
    return NeRFEngine.__new__(NeRFEngine)          # no context: the methods under test read class attributes only


def test_rebalanced_state_is_the_same_function_bit_for_bit(pkg):
    from efficient_nerf_amd.teacher import rebalanced_state
    sd = O.make_teacher_state(2)
    mx = {'h0': 2.8, 'h1': 1.9, 'h2': 0.3, 'h3': 2.7, 'h4': 700., 'h5': 5.7, 'h6': 6.8, 'h7': 31.1, 'feature': 44.7, 'views': 125.6}
    sd2, sh = rebalanced_state(sd, mx)
    assert sh == {'h0': 0, 'h1': 0, 'h2': -4, 'h3': -1, 'h4': 7, 'h5': 0, 'h6': 0, 'h7': 2, 'feature': 3, 'views': 4}
    for k, v in mx.items():
        if k not in ('h0', 'h1'):
            assert 4.0 < v / 2.0 ** sh[k] <= 8.0
    g = torch.Generator().manual_seed(0)
    pts = torch.rand((4096, 3), generator=g) * 4 - 2
    vd = torch.nn.functional.normalize(torch.randn((4096, 3), generator=g), dim=-1)
    x = torch.cat([O.nerf_embed(pts, 10), O.nerf_embed(vd, 4)], -1)
    with torch.no_grad():
        a, b = O.teacher_forward(sd, x), O.teacher_forward(sd2, x)
        a64, b64 = O.teacher_forward(sd, x, dtype=torch.float64), O.teacher_forward(sd2, x, dtype=torch.float64)
    assert torch.equal(a, b) and torch.equal(a64, b64)           # powers of two: no rounding anywhere, in float32 as in float64
    assert all(torch.equal(sd[k], O.make_teacher_state(2)[k]) for k in sd)       # the input is not modified
    # DataParallel prefixes are tolerated; a dead layer (maximum 0) keeps its scale
    sd3, sh3 = rebalanced_state({'module.' + k: v for k, v in sd.items()}, dict(mx, h3=0.0))
    assert sh3['h3'] == 0 and set(sd3) == set(sd)


def test_far_plane_ties_are_left_out_of_acc_and_depth_only(pkg):
    eng = _engine()
    n = 6
    raw_a, raw_b = torch.zeros((n, 4, 4)), torch.zeros((n, 4, 4))
    # last-sample densities: ray 0 a tie (+1e-6 / -2e-6), ray 1 the same sign, ray 2 opposite signs but far from zero, ray 3 a tie
    raw_a[:, -1, 3] = torch.tensor([1e-6, 2e-6, 0.5, -3e-7, -4., 3.])
    raw_b[:, -1, 3] = torch.tensor([-2e-6, 5e-6, -0.5, 4e-7, -4., 3.])
    tie = eng._far_ties(raw_a, raw_b)
    assert tie.tolist() == [True, False, False, True, False, False]
    got = {'rgb_map': torch.zeros((n, 3)), 'acc_map': torch.zeros(n), 'depth_map': torch.zeros(n)}
    ref = {k: v.clone() for k, v in got.items()}
    got['acc_map'][0], got['depth_map'][0] = 0.82, 4.9          # the tie's jump: left out
    got['acc_map'][2] = 3e-5                                    # not a tie: counted
    got['rgb_map'][3, 1] = 7e-5                                 # rgb is compared on every ray, ties included
    d = eng._map_diffs(got, ref, tie)
    assert d['far_plane_ties'] == 2 and abs(d['acc_map'] - 3e-5) < 1e-9 and d['depth_map'] == 0.0 and abs(d['rgb_map'] - 7e-5) < 1e-9
    d = eng._map_diffs(got, ref)
    assert abs(d['acc_map'] - 0.82) < 1e-6 and 'far_plane_ties' not in d
