"""CPU: the oracle (oracle/r2l_oracle.py) against the golden vectors generated from the
reference's own modules (tests/golden/make_golden.py).  Pins the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import r2l_oracle as O

T = torch.from_numpy


@pytest.fixture(scope='module')
def g_r2l(golden_dir):
    return np.load(os.path.join(golden_dir, 'r2l_w256d88.npz'))


@pytest.fixture(scope='module')
def g_teacher(golden_dir):
    return np.load(os.path.join(golden_dir, 'teacher_d8w256.npz'))


@pytest.fixture(scope='module')
def g_scan(golden_dir):
    return np.load(os.path.join(golden_dir, 'scan_cases.npz'))


def state_checksum(sd):
    rows = []
    for k, v in sd.items():
        f = v.double().flatten()
        idx = torch.linspace(0, f.numel() - 1, 8).long()
        rows.append(np.concatenate([[f.sum().item(), f.abs().sum().item()], f[idx].numpy()]))
    return np.stack(rows)


@pytest.fixture(scope='module')
def sd88():
    return O.make_r2l_state(0)


def test_seeded_state_matches_reference_construction(g_r2l, sd88):
    assert list(sd88.keys()) == O.r2l_state_names(43)
    assert sum(v.numel() for v in sd88.values()) == 5917187  # 23.7 MB
    np.testing.assert_array_equal(state_checksum(sd88), g_r2l['state_checksum'])


@pytest.mark.parametrize('H', [8, 400, 800])
def test_sampler_and_embedder_bit_exact(g_r2l, H):
    focal = float(g_r2l[f'focal_{H}'])
    assert focal == O.focal_from_angle(H)
    dirs = O.camera_dirs(H, H, focal)
    z = O.sampler_z_vals(16, 2., 6.)
    idx = T(g_r2l[f'idx_{H}'])
    np.testing.assert_array_equal(z.numpy(), g_r2l[f'z_vals_{H}'])
    np.testing.assert_array_equal(dirs.reshape(-1, 3)[idx].numpy(), g_r2l[f'dirs_{H}'])
    for p in range(4):
        c2w = T(g_r2l['poses'][p])
        pts = O.sample_test(dirs, z, c2w[:3, :4])[idx]
        np.testing.assert_array_equal(pts.numpy(), g_r2l[f'pts_{H}_{p}'])
        np.testing.assert_array_equal(O.positional_embed(pts[:4], 10).numpy(), g_r2l[f'emb_{H}_{p}'])
        # given-rays path
        ro, rd = O.get_rays(H, H, focal, c2w[:3, :4])
        pts2 = O.sample_rays(ro.reshape(-1, 3)[idx], rd.reshape(-1, 3)[idx], z)
        np.testing.assert_array_equal(pts2.numpy(), g_r2l[f'pts_{H}_{p}'])


def test_poses_restated(g_r2l):
    want = g_r2l['poses']
    got = torch.stack([O.pose_spherical(t, -30., 4.) for t in (-180., -37.8, 91.8)] + [O.rand_poses(3, 0)[2]])
    np.testing.assert_array_equal(got.numpy(), want)
    assert O.novel_poses(200).shape == (200, 4, 4)


@pytest.mark.parametrize('H', [8, 400, 800])
def test_r2l_forward_against_reference(g_r2l, sd88, H):
    for p in range(4):
        emb = O.positional_embed(T(g_r2l[f'pts_{H}_{p}']), 10)
        rgb = O.r2l_forward(sd88, emb)
        assert np.abs(rgb.numpy() - g_r2l[f'rgb_{H}_{p}']).max() < 2e-6


def test_r2l_layer_activations(g_r2l, sd88):
    pts = T(g_r2l['pts_400_1'][:4])
    rgb, layers = O.r2l_forward(sd88, O.positional_embed(pts, 10), return_layers=True)
    acts = g_r2l['layer_acts']
    assert acts.shape == (44, 4, 256)
    for i, l in enumerate(layers):
        assert np.abs(l.numpy() - acts[i]).max() < 1e-5, i
    assert np.abs(rgb.numpy() - g_r2l['layer_rgb']).max() < 2e-6


def test_r2l_flop_count_matches_paper():
    macs = 1008 * 256 + 86 * 256 * 256 + 256 * 3
    assert 2 * macs == 11789824  # 11.79 M FLOPs / ray


# ---------------------------------------------------------------- teacher
def test_teacher_states(g_teacher):
    for seed, key in ((1, 'state_checksum_coarse'), (2, 'state_checksum_fine')):
        sd = O.make_teacher_state(seed)
        assert list(sd.keys()) == O.teacher_state_names()
        np.testing.assert_array_equal(state_checksum(sd), g_teacher[key])
        assert sum(v.numel() for v in sd.values()) == 595844  # 2.4 MB


def test_teacher_pipeline(g_teacher):
    g = g_teacher
    sd0, sd1 = O.make_teacher_state(1), O.make_teacher_state(2)
    ro, rd = T(g['rays_o']), T(g['rays_d'])
    focal = float(g['focal'])
    ro_full, rd_full = O.get_rays(400, 400, focal, T(g['c2w'])[:3, :4])
    idx = T(g['idx'])
    np.testing.assert_array_equal(ro_full.reshape(-1, 3)[idx].numpy(), g['rays_o'])
    np.testing.assert_array_equal(rd_full.reshape(-1, 3)[idx].numpy(), g['rays_d'])
    np.testing.assert_array_equal(O.nerf_embed(T(g['pts8']), 10).numpy(), g['embedded8'][:, :63])
    for white, t in ((True, 'w'), (False, 'b')):
        o = O.render_rays(sd0, sd1, ro, rd, white_bkgd=white)
        if white:
            np.testing.assert_array_equal(o['z_vals'][:, :0].numpy().shape, (64, 0))
            assert np.abs(o['raw0'].numpy() - g['raw0']).max() < 1e-5
            assert np.abs(o['raw'].numpy() - g['raw']).max() < 1e-4
            assert np.abs(o['z_samples'].numpy() - g['z_samples']).max() < 1e-4
        for name, key in (('rgb_map', 'rgb'), ('disp_map', 'disp'), ('acc_map', 'acc'), ('depth_map', 'depth'),
                          ('rgb0', 'rgb0'), ('disp0', 'disp0'), ('acc0', 'acc0')):
            assert np.abs(o[name].numpy() - g[f'{key}_{t}']).max() < 5e-5, (name, t)
    # stage-exact pieces on the reference's own intermediate tensors
    z0, raw0 = T(g['z_vals0']), T(g['raw0'])
    w0 = O.raw2outputs(raw0, z0, rd, True)[3]
    np.testing.assert_array_equal(w0.numpy(), g['weights0'])
    zs = O.sample_pdf(T(g['z_mid']), w0[..., 1:-1], 128)
    np.testing.assert_array_equal(zs.numpy(), g['z_samples'])
    np.testing.assert_array_equal(O.merge_z(z0, zs).numpy(), g['z_all'])
    np.testing.assert_array_equal(O.coarse_z_vals(2., 6., 64, 64).numpy(), g['z_vals0'])
    r = O.raw2outputs(T(g['raw']), T(g['z_all']), rd, True)
    np.testing.assert_array_equal(r[0].numpy(), g['rgb_w'])
    np.testing.assert_array_equal(r[3].numpy(), g['weights'])


@pytest.mark.parametrize('S', [64, 192])
@pytest.mark.parametrize('white', [0, 1])
def test_raw2outputs_adversarial(g_scan, S, white):
    r = O.raw2outputs(T(g_scan[f'raw_{S}']), T(g_scan[f'z_{S}']), T(g_scan[f'rays_d_{S}']), bool(white))
    for name, val in zip(['rgb', 'disp', 'acc', 'weights', 'depth'], r):
        np.testing.assert_array_equal(val.numpy(), g_scan[f'{name}_{S}_{white}'])  # NaNs compare equal


def test_sample_pdf_adversarial(g_scan):
    zs = O.sample_pdf(T(g_scan['pdf_bins']), T(g_scan['pdf_weights']), 128)
    np.testing.assert_array_equal(zs.numpy(), g_scan['pdf_samples'])
    merged = O.merge_z(O.coarse_z_vals(2., 6., 64, zs.shape[0]), zs)
    np.testing.assert_array_equal(merged.numpy(), g_scan['pdf_merged'])
    assert (merged[:, 1:] >= merged[:, :-1]).all()


def test_oracle_ndc_rays_matches_reference(golden_dir):
    """ndc_rays (helpers:260-279) restated in the oracle == the reference's outputs, bit for bit."""
    g = np.load(os.path.join(golden_dir, 'metrics.npz'))
    ro, rd = torch.from_numpy(g['ndc_in_o']), torch.from_numpy(g['ndc_in_d'])
    for H, W, f in ((378, 504, 407.5657), (400, 400, 555.5555155968841)):
        o, d = O.ndc_rays(H, W, f, 1., ro, rd)
        assert np.array_equal(o.numpy(), g[f'ndc_o_{H}']) and np.array_equal(d.numpy(), g[f'ndc_d_{H}'])


def test_oracle_randomness_matches_reference_streams(golden_dir):
    """perturb / raw_noise_std / pytest streams of the oracle against vectors from the imported reference
    (tests/golden/make_golden_rand.py)"""
    gr = np.load(os.path.join(golden_dir, 'rand_cases.npz'))
    sc = np.load(os.path.join(golden_dir, 'scan_cases.npz'))
    bins, w = torch.from_numpy(sc['pdf_bins']), torch.from_numpy(sc['pdf_weights'])
    s, cdf, inds = O.sample_pdf(bins, w, 128, det=True, taps=True)
    assert np.array_equal(cdf.numpy(), gr['pdf_cdf']) and np.array_equal(inds.numpy(), gr['pdf_inds'])
    for det in (False, True):
        assert np.array_equal(O.sample_pdf(bins, w, 128, det=det, pytest=True).numpy(), gr[f'pdf_samples_pytest_det{int(det)}'])
    for S in (64, 192):
        raw, z, rd = (torch.from_numpy(sc[f'{k}_{S}']) for k in ('raw', 'z', 'rays_d'))
        r = O.raw2outputs(raw, z, rd, True, raw_noise_std=0.7, pytest=True)
        assert np.array_equal(r[3].numpy(), gr[f'noise_weights_{S}_1'], equal_nan=True)
    zc = O.perturb_z_vals(O.coarse_z_vals(2., 6., 64, gr['rr_z_coarse'].shape[0]), pytest=True)
    assert np.array_equal(zc.numpy(), gr['rr_z_coarse'])


def test_variants_of_the_constructor_match_the_reference():
    """tests/golden/r2l_variants.npz: the reference's own NeRF_v3_2 with other activations, res_scale and the plain-MLP body
    (generated by make_golden_variants.py from the imported reference class); the oracle rebuilds each seeded network -- RNG draw
    for RNG draw -- and must reproduce the reference's rgb"""
    import os
    import numpy as np
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'r2l_variants.npz'))
    H, focal = int(g['H']), float(g['focal'])
    c2w = torch.from_numpy(g['c2w'])
    pts = O.sample_test(O.camera_dirs(H, H, focal), O.sampler_z_vals(16, 2., 6.), c2w[:3, :4])
    emb = O.positional_embed(pts[torch.from_numpy(g['idx'])], 10)
    names = [k[:-4] for k in g.files if k.endswith('_cfg')]
    assert len(names) == 5
    for name in names:
        _, D, arch, act, inact, outact, rs, seed = [str(x) for x in g[name + '_cfg']]
        if arch == 'mlp':
            out = O.r2l_forward_mlp(O.make_r2l_mlp_state(int(seed), netdepth=int(D)), emb, act=act)
        else:
            out = O.r2l_forward(O.make_r2l_state(int(seed), netdepth=int(D), inact=inact), emb, res_scale=float(rs), act=act, inact=inact,
                                outact=outact)
        assert (out - torch.from_numpy(g[name + '_rgb'])).abs().max().item() <= 2e-6, name
