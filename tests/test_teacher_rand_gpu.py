"""GPU: the teacher path's integer work and the reference's training-time randomness, against vectors generated
from the imported reference (tests/golden/make_golden_rand.py):
  * sample_pdf taps: cdf bit-exact, searchsorted indices exact, samples bit-exact (utils/run_nerf_raybased_helpers.py:283-330);
  * sample_pdf with the pytest numpy streams (det False / True);
  * raw2outputs with raw_noise_std > 0 (main.py:592-600);
  * render_rays with perturb = 1 + noise (main.py:684-699) and its full return set rgb0 / disp0 / acc0 / z_std
    (main.py:743-750);
  * a full 400 x 400 teacher frame: finite, in range, deterministic, equal to its row-range renders, <= 1e-4 vs the CPU
    oracle on a strided 2,000-ray subset."""
import os

import numpy as np
import pytest
import torch

from oracle import r2l_oracle as O

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.fixture(scope='module')
def gr(golden_dir):
    return np.load(os.path.join(golden_dir, 'rand_cases.npz'))


@pytest.fixture(scope='module')
def gs(golden_dir):
    return np.load(os.path.join(golden_dir, 'scan_cases.npz'))


@pytest.fixture(scope='module')
def g(golden_dir):
    return np.load(os.path.join(golden_dir, 'teacher_d8w256.npz'))


@pytest.fixture(scope='module')
def engine(pkg, g):
    from efficient_nerf_amd import NeRFEngine
    eng = NeRFEngine(400, 400, float(g['focal'])).load_state_dicts(O.make_teacher_state(1), O.make_teacher_state(2))
    yield eng
    eng.close()


def test_sample_pdf_integer_work_is_pinned(pkg, gs, gr):
    from efficient_nerf_amd import sample_pdf
    bins, w = T(gs['pdf_bins']).cuda(), T(gs['pdf_weights']).cuda()
    zs, cdf, inds = sample_pdf(bins, w, 128, det=True, taps=True)
    n_cdf = int((cdf.cpu().numpy() != gr['pdf_cdf']).sum())
    n_ind = int((inds.cpu().numpy() != gr['pdf_inds']).sum())
    n_s = int((zs.cpu().numpy() != gs['pdf_samples']).sum())
    print(f'sample_pdf vs reference golden: {n_cdf} cdf values, {n_ind} indices, {n_s} samples differ of {zs.numel()}')
    assert n_cdf == 0 and n_ind == 0 and n_s == 0
    # and, independent of any golden: the indices ARE searchsorted(cdf, u, right=True) of the kernel's own cdf
    u = torch.linspace(0., 1., 128).expand(cdf.shape[0], 128).contiguous()
    assert torch.equal(inds.cpu().long(), torch.searchsorted(cdf.cpu(), u, right=True))
    # u evaluated in the kernel (nerf_sample_pdf, no u argument: scalar linspace formula; the host's vectorised
    # torch.linspace may differ in the last ulp of u)
    from efficient_nerf_amd import _lib
    out = torch.empty_like(zs)
    _lib.check(_lib.lib().nerf_sample_pdf(_lib.dptr(bins), _lib.dptr(w), bins.shape[0], bins.shape[1], 128, _lib.dptr(out),
                                          _lib.current_stream()))
    assert float((out - zs).abs().max()) <= 1e-5


@pytest.mark.parametrize('det', [False, True])
def test_sample_pdf_pytest_streams(pkg, gs, gr, det):
    from efficient_nerf_amd import sample_pdf
    bins, w = T(gs['pdf_bins']).cuda(), T(gs['pdf_weights']).cuda()
    zs = sample_pdf(bins, w, 128, det=det, pytest=True)
    np.testing.assert_array_equal(zs.cpu().numpy(), gr[f'pdf_samples_pytest_det{int(det)}'])


@pytest.mark.parametrize('S,white', [(64, 0), (64, 1), (192, 0), (192, 1)])
def test_raw2outputs_noise(pkg, gs, gr, S, white):
    from efficient_nerf_amd import raw2outputs
    raw, z, rd = (T(gs[f'{k}_{S}']).cuda() for k in ('raw', 'z', 'rays_d'))
    out = raw2outputs(raw, z, rd, 0.7, bool(white), pytest=True)
    for name, val in zip(['rgb', 'disp', 'acc', 'weights', 'depth'], out):
        want, got = gr[f'noise_{name}_{S}_{white}'], val.cpu().numpy()
        assert (np.isnan(got) == np.isnan(want)).all()
        m = ~np.isnan(want)
        if name == 'disp':
            assert (np.abs(got[m] - want[m]) <= 2e-5 * np.abs(want[m]) + 1e-6).all()
        else:
            assert np.abs(got[m] - want[m]).max() <= 3e-6 * max(1., np.abs(want[m]).max())
    # the noise matters: without it the weights differ
    w0 = raw2outputs(raw, z, rd, 0., bool(white))[3].cpu().numpy()
    assert np.nanmax(np.abs(w0 - gr[f'noise_weights_{S}_{white}'])) > 1e-3


def test_render_rays_perturb_and_noise(engine, g, gr):
    ro, rd = T(g['rays_o']).cuda(), T(g['rays_d']).cuda()
    out = engine.render_rays(ro, rd, extras=True, perturb=1., raw_noise_std=0.5, pytest=True)
    z = out['z_vals'].cpu().numpy()
    # the jittered coarse depths are among the merged depths, bit for bit
    zc = gr['rr_z_coarse']
    assert all(np.isin(zc[i], z[i]).all() for i in range(zc.shape[0]))
    assert np.abs(out['z_samples'].cpu().numpy() - gr['rr_z_samples']).max() <= 1e-4
    for k, tol in (('rgb0', 2e-5), ('acc0', 2e-5), ('z_std', 2e-5)):
        assert np.abs(out[k].cpu().numpy() - gr[f'rr_{k}']).max() <= tol, k
    for k, name in (('rgb_map', 'rgb'), ('acc_map', 'acc'), ('depth_map', 'depth')):
        assert np.abs(out[k].cpu().numpy() - gr[f'rr_{name}']).max() <= 1e-4, k
    d, dw = out['disp0'].cpu().numpy(), gr['rr_disp0']
    assert (np.abs(d - dw) <= 1e-4 * np.abs(dw) + 1e-6).all()


def test_render_rays_return_set(engine, g, gr):
    """main.py:743-750 on the deterministic test path: rgb0, disp0, acc0, z_std"""
    ro, rd = T(g['rays_o']).cuda(), T(g['rays_d']).cuda()
    out = engine.render_rays(ro, rd, extras=True)
    assert np.abs(out['rgb0'].cpu().numpy() - g['rgb0_w']).max() <= 1e-5
    assert np.abs(out['acc0'].cpu().numpy() - g['acc0_w']).max() <= 1e-5
    d, dw = out['disp0'].cpu().numpy(), g['disp0_w']
    assert (np.abs(d - dw) <= 2e-5 * np.abs(dw) + 1e-6).all()
    assert np.abs(out['z_samples'].cpu().numpy() - g['z_samples']).max() <= 1e-5  # (the coarse weights come from the HIP MLP)
    assert np.abs(out['z_std'].cpu().numpy() - gr['det_z_std']).max() <= 1e-6


@pytest.mark.parametrize('prec', ['fp16x3', 'fp16_fp8', 'fp16x1', 'fp16x3_asm'])
def test_full_teacher_frame(engine, g, prec):
    from efficient_nerf_amd import PRECISIONS
    engine.set_precision(PRECISIONS[prec])
    c2w = T(g['c2w'])
    H = W = 400
    full = engine.render(c2w)
    rgb = full['rgb_map']
    assert rgb.shape == (H * W, 3) and bool(torch.isfinite(rgb).all())
    assert float(rgb.min()) >= -1e-6 and float(rgb.max()) <= 1. + 1e-5
    assert bool(torch.isfinite(full['acc_map']).all()) and float(full['acc_map'].max()) <= 1. + 1e-5
    again = engine.render(c2w)
    assert all(torch.equal(full[k], again[k]) for k in ('rgb_map', 'acc_map', 'depth_map'))
    # row ranges of the frame == the frame
    for r0, r1 in ((0, 37), (37, 211), (211, 400)):
        part = engine.render(c2w, rows=(r0, r1))
        assert torch.equal(part['rgb_map'], rgb[r0 * W:r1 * W]), (r0, r1)
    # CPU oracle on a strided subset of 2,000 rays
    idx = torch.arange(0, H * W, 80)[:2000]       # (the whole frame of fp16x1 against fp16x3: tests/test_teacher_gpu.py)
    ro, rd = O.get_rays(H, W, float(g['focal']), c2w[:3, :4])
    ref = O.render_rays(O.make_teacher_state(1), O.make_teacher_state(2), ro.reshape(-1, 3)[idx].float(),
                        rd.reshape(-1, 3)[idx].float(), white_bkgd=True)
    err = (rgb.cpu()[idx] - ref['rgb_map']).abs().max().item()
    print(f'teacher full frame {prec}: L_inf vs CPU oracle on 2,000 rays {err:.2e}')
    assert err <= 1e-4
    engine.set_precision(PRECISIONS['fp16x3'])
