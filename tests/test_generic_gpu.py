"""GPU parity of the generic fp32 layer path (csrc/r2l_generic.hip through the C-ABI, composed by efficient-nerf_amd/generic.py):
the layer op against torch's F.linear on the same inputs, the sampler / embedders against the oracle, and ten NeRF_v3_2
variants the fused kernels refuse against golden vectors from the reference's own classes.

Tolerances: points bit-exact (one rounding per op, as the reference); embeddings <= 5e-7; a layer <= 2e-5 x max|y| (fp32
products, fp32 accumulation: only the summation order differs from the CPU's); rgb <= 1e-4 (BASELINE.json's contract; measured ~1e-6)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import r2l_oracle as O

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.fixture(scope='module')
def g(golden_dir):
    return np.load(os.path.join(golden_dir, 'r2l_generic.npz'))


def ref_layer(x, w, b, act, res, rs, post):
    v = F.linear(x, w, b)
    if res is not None:
        v = v.mul(rs) + res
    v = {'none': lambda t: t, 'relu': F.relu, 'lrelu': F.leaky_relu, 'sigmoid': torch.sigmoid}[act](v)
    return v if post is None else v + post


@pytest.mark.parametrize('n,in_dim,out_dim,act,res,post', [
    (1, 1, 1, 'none', False, False), (5, 63, 256, 'relu', False, False), (129, 319, 256, 'relu', True, False),
    (1000, 1008, 181, 'lrelu', True, True), (4097, 256, 3, 'sigmoid', False, False), (300, 33, 65, 'none', True, True),
    (128, 32, 64, 'relu', False, True), (70000, 96, 96, 'relu', True, False)])
def test_linear_layer_matches_torch(pkg, n, in_dim, out_dim, act, res, post):
    from efficient_nerf_amd.generic import Linear
    gen = torch.Generator().manual_seed(n + in_dim)
    x = torch.randn(n, in_dim, generator=gen)
    w = (torch.rand(out_dim, in_dim, generator=gen) * 2 - 1) / in_dim ** 0.5
    b = torch.rand(out_dim, generator=gen) - 0.5
    r = torch.randn(n, out_dim, generator=gen) if res else None
    p = torch.randn(n, out_dim, generator=gen) if post else None
    want = ref_layer(x, w, b, act, r, 0.3, p)
    lin = Linear(w, b)
    # strided views: x a column slice of a wider buffer, y written into one (how the reference's torch.cat inputs are formed)
    xb = torch.full((n, in_dim + 7), 9.0, device='cuda'); xb[:, 3:3 + in_dim] = x.cuda()
    yb = torch.full((n, out_dim + 5), -7.0, device='cuda')
    got = lin(xb[:, 3:3 + in_dim], yb[:, 2:2 + out_dim], act=act, res=None if r is None else r.cuda(), res_scale=0.3,
              post=None if p is None else p.cuda())
    torch.cuda.synchronize()
    scale = max(1.0, want.abs().max().item())
    assert (got.cpu() - want).abs().max().item() <= 2e-5 * scale
    assert torch.all(yb[:, :2] == -7.0) and torch.all(yb[:, 2 + out_dim:] == -7.0)      # nothing outside the view
    # res aliasing y (a residual stream updated in place)
    if r is not None:
        y2 = r.cuda().clone()
        lin(x.cuda(), y2, act=act, res=y2, res_scale=0.3, post=None if p is None else p.cuda())
        assert torch.equal(y2, got.contiguous())


def test_layer_refuses_bad_arguments(pkg):
    from efficient_nerf_amd import R2LError
    from efficient_nerf_amd.generic import Linear
    lin = Linear(torch.zeros(4, 6), None)
    x = torch.zeros(10, 6, device='cuda')
    with pytest.raises(R2LError):
        lin(x, torch.zeros(10, 5, device='cuda'))               # wrong width
    with pytest.raises(R2LError):
        lin(x, torch.zeros(9, 4, device='cuda'))                # wrong rows
    with pytest.raises(R2LError):
        lin(x, x[:, :4])                                        # x and y overlap
    with pytest.raises(R2LError):
        lin(x.double(), torch.zeros(10, 4, device='cuda'))
    y = lin(x, torch.empty(10, 4, device='cuda'))               # no bias: zeros
    assert torch.all(y == 0)


def test_sample_points_and_embedders(pkg):
    import ctypes as C
    from efficient_nerf_amd._lib import lib, check, dptr, current_stream
    H, ns = 12, 7
    focal = O.focal_from_angle(H)
    c2w = O.pose_spherical(33., -20., 4.)
    ro, rd = O.get_rays(H, H, focal, c2w[:3, :4])
    ro, rd = ro.reshape(-1, 3).contiguous(), rd.reshape(-1, 3).contiguous()
    z = O.sampler_z_vals(ns, 2., 6.)
    pts = torch.empty(H * H, 3 * ns, device='cuda')
    rod, rdd, zd = ro.cuda(), rd.cuda(), z.cuda()               # kept alive: the library borrows the pointers
    check(lib().r2l_sample_points(dptr(rod), dptr(rdd), H * H, dptr(zd), ns, 0, dptr(pts), current_stream()))
    want = O.sample_rays(ro, rd, z)
    assert torch.equal(pts.cpu(), want.reshape(H * H, -1))
    zr = (torch.rand(H * H, ns) * 4 + 2).contiguous()           # per-ray z (main.py:701)
    zrd = zr.cuda()
    check(lib().r2l_sample_points(dptr(rod), dptr(rdd), H * H, dptr(zrd), ns, 1, dptr(pts), current_stream()))
    want = ro[:, None, :] + rd[:, None, :] * zr[:, :, None]
    assert torch.equal(pts.cpu().reshape(H * H, ns, 3), want)
    # Embedder.embed (teacher ordering), into a column slice of a wider buffer
    x = (torch.rand(500, 3) * 8 - 4)
    for L in (0, 4, 10):
        out = torch.full((500, 3 * (2 * L + 1) + 4), 5.0, device='cuda')
        xd = x.cuda()
        check(lib().nerf_embed(dptr(xd), 3, 500, 3, L, C.c_void_p(out.data_ptr() + 8), out.stride(0), current_stream()))
        torch.cuda.synchronize()
        assert (out[:, 2:2 + 3 * (2 * L + 1)].cpu() - O.nerf_embed(x, L)).abs().max().item() <= 5e-7
        assert torch.equal(out[:, 2:5].cpu(), x) and torch.all(out[:, :2] == 5.0) and torch.all(out[:, 2 + 3 * (2 * L + 1):] == 5.0)


def test_reference_variants_render_within_contract(g, pkg):
    from efficient_nerf_amd.generic import GenericR2L
    H, focal, c2w, idx = int(g['H']), float(g['focal']), T(g['c2w']), T(g['idx'])
    for cs in json.loads(str(g['cases'])):
        ns, L = cs.get('n_sample', 16), cs.get('L', 10)
        eng = GenericR2L(H, H, focal, 2., 6., n_sample=ns, L=L, netdepth=cs['netdepth'], netwidth=cs['netwidth'],
                         layerwise_netwidths=cs.get('layerwise_netwidths', ''), act=cs.get('act', 'relu'),
                         use_residual=cs.get('use_residual', True), trial=cs['trial'], z_vals=O.sampler_z_vals(ns, 2., 6.))
        sd = O.make_v3_2_state(int(g[cs['name'] + '_seed']), cs['netdepth'], cs['netwidth'], eng.input_dim, cs.get('layerwise_netwidths', ''),
                               cs.get('act', 'relu'), cs['trial'])
        eng.load_state_dict({'module.' + k: v for k, v in sd.items()})         # DataParallel prefixes tolerated
        rgb = eng.render(c2w)
        err = (rgb.cpu()[idx] - T(g[cs['name'] + '_rgb'])).abs().max().item()
        assert err <= 1e-4, (cs['name'], err)
        print(f"{cs['name']}: L_inf vs the reference {err:.1e}")
        # row ranges, chunking and the given-rays entry give the same values bit for bit
        eng.chunk = 3 * H
        part = eng.render(c2w, rows=(3, 11))
        assert torch.equal(part, rgb[3 * H:11 * H])
        ro, rd = O.get_rays(H, H, focal, c2w[:3, :4])
        rr = eng.render_rays(ro.reshape(-1, 3).cuda(), rd.reshape(-1, 3).cuda())
        assert torch.equal(rr, rgb)
        both = eng.render_batch(torch.stack([c2w[:3, :4], c2w[:3, :4]]).cuda(), rows=(0, 4))
        assert torch.equal(both[1], rgb[:4 * H]) and torch.equal(both[0], both[1])


def test_reference_nerf_variants_render_within_contract(golden_dir, pkg):
    """GenericNeRF against render_rays composed from the reference's own modules (tests/golden/make_golden_generic_nerf.py):
    other depths / widths (also of the fine network), multires, i_embed = -1, no view directions, N_importance = 0, lindisp."""
    from efficient_nerf_amd.generic import GenericNeRF
    g = np.load(os.path.join(golden_dir, 'nerf_generic.npz'))
    H, focal = int(g['H']), float(g['focal'])
    ro, rd = T(g['rays_o']).cuda(), T(g['rays_d']).cuda()
    for ci, cs in enumerate(json.loads(str(g['cases']))):
        use_vd, i_embed, Ni = cs.get('use_viewdirs', True), cs.get('i_embed', 0), cs['N_importance']
        eng = GenericNeRF(H, H, focal, 2., 6., N_samples=cs['N_samples'], N_importance=Ni, multires=cs.get('multires', 10),
                          multires_views=cs.get('multires_views', 4), i_embed=i_embed, netdepth=cs['netdepth'], netwidth=cs['netwidth'],
                          netdepth_fine=cs.get('netdepth_fine', cs['netdepth']), netwidth_fine=cs.get('netwidth_fine', cs['netwidth']),
                          use_viewdirs=use_vd, white_bkgd=cs.get('white_bkgd', True), lindisp=cs.get('lindisp', False))
        seed = int(g[cs['name'] + '_seed'])
        mk = lambda s, D, W: O.make_nerf_state(s, D, W, eng.input_ch, eng.input_ch_views, eng.output_ch, (4,), use_vd)
        sd0 = mk(seed, cs['netdepth'], cs['netwidth'])
        sd1 = mk(seed + 1, cs.get('netdepth_fine', cs['netdepth']), cs.get('netwidth_fine', cs['netwidth'])) if Ni > 0 else None
        eng.load_state_dicts(sd0, sd1)
        out = eng.render_rays(ro, rd, extras=True)
        worst = {}
        for k, tol in (('rgb_map', 1e-4), ('acc_map', 1e-4), ('depth_map', 1e-3), ('raw', 2e-4), ('z_samples', 1e-3), ('rgb0', 1e-4)):
            if f"{cs['name']}_{k}" not in g.files:
                continue
            err = (out[k].cpu() - T(g[f"{cs['name']}_{k}"])).abs().max().item()
            worst[k] = err
            assert err <= tol, (cs['name'], k, err)
        dref = T(g[cs['name'] + '_disp_map'])
        assert ((out['disp_map'].cpu() - dref).abs() / dref.abs().clamp_min(1e-6)).max().item() <= 1e-3, cs['name']
        print(f"{cs['name']}: " + ', '.join(f'{k} {v:.1e}' for k, v in worst.items()))
        # chunking does not change a value; the pose entry is the rays entry on the same rays
        eng.chunk = 7
        again = eng.render_rays(ro, rd)
        assert torch.equal(again['rgb_map'], out['rgb_map'])
    frame = eng.render(T(g['c2w']), rows=(3, 5))
    idx = T(g['idx'])
    sel = [(int(i), k) for k, i in enumerate(idx) if 3 * H <= int(i) < 5 * H]
    for i, k in sel:
        assert torch.equal(frame['rgb_map'][i - 3 * H], out['rgb_map'][k])


def test_mirror_call_chain_on_a_generic_shape(g, pkg):
    """model(positional_embedder(point_sampler.sample_test(c2w))) (main.py:300-309) with the reference's own constructor
    arguments for a network the fused kernels refuse: the handles are consumed on the generic path."""
    from types import SimpleNamespace
    from efficient_nerf_amd import NeRF_v3_2, PointSampler, PositionalEmbedder, render_func
    cs = [c for c in json.loads(str(g['cases'])) if c['name'] == 'w181_d10_ns8_L6'][0]
    H, focal = int(g['H']), float(g['focal'])
    args = SimpleNamespace(netdepth=cs['netdepth'], netwidth=cs['netwidth'], layerwise_netwidths='', act='relu', linear_tail=False,
                           use_residual=True, trial=SimpleNamespace(**cs['trial']))
    input_dim = 3 * 8 * 13
    sd = O.make_v3_2_state(int(g[cs['name'] + '_seed']), cs['netdepth'], cs['netwidth'], input_dim, '', 'relu', cs['trial'])
    model = NeRF_v3_2(args, input_dim, 3).load_state_dict(sd)
    ps, pe = PointSampler(H, H, focal, 8, 2., 6.), PositionalEmbedder(L=6)
    rgb = render_func(model, T(g['c2w'])[:3, :4], ps, pe)
    assert (rgb.cpu()[T(g['idx'])] - T(g[cs['name'] + '_rgb'])).abs().max().item() <= 1e-4
    rays = O.get_rays(H, H, focal, T(g['c2w'])[:3, :4])
    rgb2 = model(pe(ps.sample_train(rays[0].reshape(-1, 3).cuda(), rays[1].reshape(-1, 3).cuda(), perturb=0)))
    assert torch.equal(rgb, rgb2)
    with pytest.raises(Exception):
        model(pe(PointSampler(H, H, focal, 16, 2., 6.).sample_test(T(g['c2w'])[:3, :4])))     # 1008 features into a 312-input network


def test_linear_layer_random_shapes(pkg):
    """24 seeded shapes (primes, 1-wide, just past the 128 x 64 x 32 tile edges) against F.linear in float64"""
    from efficient_nerf_amd.generic import Linear
    rng = np.random.RandomState(7)
    edge = [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 257, 1009]
    for case in range(24):
        n, i, o = int(rng.choice(edge + [3001])), int(rng.choice(edge)), int(rng.choice(edge))
        gen = torch.Generator().manual_seed(case)
        x, w, b = torch.randn(n, i, generator=gen), torch.randn(o, i, generator=gen) / i ** 0.5, torch.randn(o, generator=gen)
        act = ('none', 'relu', 'lrelu', 'sigmoid')[case % 4]
        want = ref_layer(x.double(), w.double(), b.double(), act, None, 1.0, None)
        got = Linear(w, b)(x.cuda(), torch.empty(n, o, device='cuda'), act=act)
        assert (got.cpu().double() - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item()), (n, i, o, act)
