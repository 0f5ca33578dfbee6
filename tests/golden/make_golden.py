"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own Python
modules (model/nerf_raybased.py, utils/run_nerf_raybased_helpers.py) on CPU.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The reference never travels; only the .npz files written here do.  Each fixture holds
inputs and the reference's outputs.  While generating, the script also asserts that the
repo's oracle (oracle/r2l_oracle.py) reproduces every vector, so a drifted oracle fails
here before it fails in tests/.
"""
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
import utils.run_nerf_raybased_helpers as RH  # noqa: E402  (reference)
from oracle import r2l_oracle as O  # noqa: E402

torch.set_grad_enabled(False)
torch.autograd.set_detect_anomaly(False)


def r2l_args():
    return SimpleNamespace(netdepth=88, netwidth=256, layerwise_netwidths='', act='relu',
                           linear_tail=False, use_residual=True,
                           trial=SimpleNamespace(body_arch='resmlp', n_block=-1, n_learnable=2,
                                                 res_scale=1., inact='relu', outact='none'))


def state_checksum(sd):
    """float64 sum, abs-sum and 8 sampled values per tensor (cheap identity check)."""
    rows = []
    for k, v in sd.items():
        f = v.double().flatten()
        idx = torch.linspace(0, f.numel() - 1, 8).long()
        rows.append(np.concatenate([[f.sum().item(), f.abs().sum().item()], f[idx].numpy()]))
    return np.stack(rows)


def test_poses():
    thetas = [-180., -37.8, 91.8]
    poses = [O.pose_spherical(t, -30., 4.) for t in thetas]
    poses.append(O.rand_poses(3, seed=0)[2])
    return torch.stack(poses, 0)


def gen_r2l():
    torch.manual_seed(0)
    model = RM.NeRF_v3_2(r2l_args(), 1008, 3).eval()
    sd_ref = {k: v.clone() for k, v in model.state_dict().items()}
    sd = O.make_r2l_state(0)
    assert list(sd.keys()) == list(sd_ref.keys()) == O.r2l_state_names()
    for k in sd:
        assert torch.equal(sd[k], sd_ref[k]), k
    poses = test_poses()
    out = dict(poses=poses.numpy(), state_checksum=state_checksum(sd_ref),
               near=np.float32(2.), far=np.float32(6.))
    pe = RM.PositionalEmbedder(L=10)
    for (H, W) in [(400, 400), (800, 800), (8, 8)]:
        focal = O.focal_from_angle(W)
        ps = RM.PointSampler(H, W, focal, 16, 2., 6.)
        n = H * W
        idx = torch.arange(0, n, max(1, n // 256))[:256]
        tag = f'{H}'
        out[f'focal_{tag}'] = np.float64(focal)
        out[f'idx_{tag}'] = idx.numpy()
        out[f'z_vals_{tag}'] = ps.z_vals.numpy()
        assert torch.equal(ps.dirs, O.camera_dirs(H, W, focal))
        assert torch.equal(ps.z_vals, O.sampler_z_vals(16, 2., 6.))
        out[f'dirs_{tag}'] = ps.dirs.reshape(-1, 3)[idx].numpy()
        for p, c2w in enumerate(poses):
            pts = ps.sample_test(c2w[:3, :4])  # [H*W, 48]
            assert torch.equal(pts, O.sample_test(ps.dirs, ps.z_vals, c2w[:3, :4]))
            sub = pts[idx]
            emb = pe(sub)
            assert torch.equal(emb, O.positional_embed(sub, 10))
            rgb = model(emb)
            rgb_o = O.r2l_forward(sd, emb)
            assert (rgb - rgb_o).abs().max() < 1e-6, (rgb - rgb_o).abs().max()
            out[f'pts_{tag}_{p}'] = sub.numpy()
            out[f'emb_{tag}_{p}'] = emb[:4].numpy()
            out[f'rgb_{tag}_{p}'] = rgb.numpy()
            # given-rays path (main.py:220-223): sample_train(rays_o, rays_d, perturb=0)
            ro, rd = RH.get_rays(H, W, focal, c2w[:3, :4])
            ro2, rd2 = O.get_rays(H, W, focal, c2w[:3, :4])
            assert torch.equal(ro, ro2) and torch.equal(rd, rd2)
            pts2 = ps.sample_train(ro.reshape(-1, 3)[idx], rd.reshape(-1, 3)[idx], perturb=0)
            assert torch.equal(pts2, sub)
            if tag == '400':
                out[f'rays_d_{tag}_{p}'] = rd.reshape(-1, 3)[idx].numpy()
    # per-layer activations for 4 rays (pose 1, 400x400), hooks on the reference modules
    focal = O.focal_from_angle(400)
    ps = RM.PointSampler(400, 400, focal, 16, 2., 6.)
    emb = pe(ps.sample_test(poses[1][:3, :4])[out['idx_400'][:4]])
    acts = []
    hooks = [model.head.register_forward_hook(lambda m, i, o: acts.append(o.clone()))]
    for blk in model.body:
        hooks.append(blk.register_forward_hook(lambda m, i, o: acts.append(o.clone())))
    rgb = model(emb)
    for h in hooks:
        h.remove()
    _, layers = O.r2l_forward(sd, emb, return_layers=True)
    assert len(layers) == len(acts) == 44
    for a, b in zip(acts, layers):
        assert (a - b).abs().max() < 1e-5
    out['layer_acts'] = torch.stack(acts, 0).numpy()  # [44, 4, 256]
    out['layer_rgb'] = rgb.numpy()
    np.savez_compressed(os.path.join(HERE, 'r2l_w256d88.npz'), **out)
    print('r2l_w256d88.npz', {k: v.shape for k, v in out.items() if hasattr(v, 'shape')}.__len__(), 'arrays')


def gen_teacher():
    H = W = 400
    focal = O.focal_from_angle(W)
    nets, sds = [], []
    for seed in (1, 2):
        torch.manual_seed(seed)
        net = RM.NeRF(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27,
                      use_viewdirs=True).eval()
        net.alpha_linear.bias.data += 0.5
        sd = O.make_teacher_state(seed)
        for k, v in net.state_dict().items():
            assert torch.equal(v, sd[k]), k
        assert list(net.state_dict().keys()) == list(sd.keys()) == O.teacher_state_names()
        nets.append(net)
        sds.append(sd)
    out = dict(state_checksum_coarse=state_checksum(nets[0].state_dict()),
               state_checksum_fine=state_checksum(nets[1].state_dict()), focal=np.float64(focal))
    embed_fn, ch = RH.get_embedder(10, 0)
    embeddirs_fn, chv = RH.get_embedder(4, 0)
    assert ch == 63 and chv == 27
    c2w = O.pose_spherical(-37.8, -30., 4.)
    out['c2w'] = c2w.numpy()
    rays_o, rays_d = RH.get_rays(H, W, focal, c2w[:3, :4])
    idx = torch.arange(0, H * W, H * W // 64)[:64] + 137
    rays_o, rays_d = rays_o.reshape(-1, 3)[idx].float(), rays_d.reshape(-1, 3)[idx].float()
    out['idx'] = idx.numpy()
    out['rays_o'], out['rays_d'] = rays_o.numpy(), rays_d.numpy()
    # --- main.py:148-175 ray packing (restated: main.py is not importable) from reference pieces
    viewdirs = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    out['viewdirs'] = viewdirs.numpy()
    for white in (True, False):
        # --- main.py:676-756 render_rays composed from the reference's importable functions
        near, far = 2. * torch.ones_like(rays_d[..., :1]), 6. * torch.ones_like(rays_d[..., :1])
        t_vals = torch.linspace(0., 1., steps=64)
        z_vals = (near * (1. - t_vals) + far * t_vals).expand([64, 64])
        pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]
        raw0 = RH.run_network(pts, viewdirs, nets[0], embed_fn, embeddirs_fn, netchunk=1024 * 64)
        rgb0, disp0, acc0, w0, depth0 = RH.raw2outputs(raw0, z_vals, rays_d, 0, white)
        rgb0m, disp0m, acc0m, w0m, depth0m = RM.raw2outputs(raw0, z_vals, rays_d, 0, white)
        assert torch.equal(rgb0, rgb0m) and torch.equal(w0, w0m)
        z_mid = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
        z_samples = RH.sample_pdf(z_mid, w0[..., 1:-1], 128, det=True)
        z_all, _ = torch.sort(torch.cat([z_vals, z_samples], -1), -1)
        pts = rays_o[..., None, :] + rays_d[..., None, :] * z_all[..., :, None]
        raw = RH.run_network(pts, viewdirs, nets[1], embed_fn, embeddirs_fn, netchunk=1024 * 64)
        rgb, disp, acc, w, depth = RH.raw2outputs(raw, z_all, rays_d, 0, white)
        # oracle must agree
        o = O.render_rays(sds[0], sds[1], rays_o, rays_d, white_bkgd=white)
        assert (o['raw0'] - raw0).abs().max() < 1e-5
        assert torch.equal(O.raw2outputs(raw0, z_vals, rays_d, white)[3], w0)
        assert torch.equal(O.sample_pdf(z_mid, w0[..., 1:-1], 128), z_samples)
        assert torch.equal(O.merge_z(z_vals, z_samples), z_all)
        assert (o['rgb_map'] - rgb).abs().max() < 2e-5, (o['rgb_map'] - rgb).abs().max()
        t = 'w' if white else 'b'
        if white:
            out['z_vals0'] = z_vals.numpy()
            out['raw0'] = raw0.numpy()
            out['z_mid'] = z_mid.numpy()
            out['z_samples'] = z_samples.numpy()
            out['z_all'] = z_all.numpy()
            out['raw'] = raw.numpy()
            out['weights0'] = w0.numpy()
            out['weights'] = w.numpy()
            emb = torch.cat([embed_fn(pts.reshape(-1, 3)[:8]),
                             embeddirs_fn(viewdirs[:1].expand(8, 3))], -1)
            assert torch.equal(emb, torch.cat([O.nerf_embed(pts.reshape(-1, 3)[:8], 10),
                                               O.nerf_embed(viewdirs[:1].expand(8, 3), 4)], -1))
            out['embedded8'] = emb.numpy()
            out['pts8'] = pts.reshape(-1, 3)[:8].numpy()
        for name, val in [('rgb0', rgb0), ('disp0', disp0), ('acc0', acc0), ('depth0', depth0),
                          ('rgb', rgb), ('disp', disp), ('acc', acc), ('depth', depth)]:
            out[f'{name}_{t}'] = val.numpy()
    np.savez_compressed(os.path.join(HERE, 'teacher_d8w256.npz'), **out)
    print('teacher_d8w256.npz', len(out), 'arrays')


def gen_scan_cases():
    """Adversarial standalone raw2outputs / sample_pdf vectors."""
    g = torch.Generator().manual_seed(7)
    out = {}
    for S in (64, 192):
        n = 48
        raw = torch.randn(n, S, 4, generator=g) * 2.
        z = torch.sort(2. + 4. * torch.rand(n, S, generator=g), -1)[0]
        if S == 64:
            z = O.coarse_z_vals(2., 6., 64, n).clone()
        rays_d = torch.randn(n, 3, generator=g)
        raw[0, :, 3] = 0.  # all-zero sigma
        raw[1, :, 3] = -5.  # relu kills everything
        raw[2, :, 3] = 1e4  # huge sigma: first sample takes all
        raw[3, :, 3] = 1e4
        raw[3, :S // 2, 3] = 0.  # empty then wall
        raw[4, :, :3] = 30.  # saturated sigmoid
        raw[5, :, :3] = -30.
        z[6] = z[6, :1].expand(S)  # zero-length intervals
        rays_d[7] = 0.  # zero direction -> dists 0 (and 1e10*0)
        out[f'raw_{S}'], out[f'z_{S}'], out[f'rays_d_{S}'] = raw.numpy(), z.numpy(), rays_d.numpy()
        for white in (False, True):
            r = RH.raw2outputs(raw, z, rays_d, 0, white)
            r2 = O.raw2outputs(raw, z, rays_d, white)
            for a, b in zip(r, r2):
                assert torch.equal(a, b) or (torch.isnan(a) == torch.isnan(b)).all()
            for name, val in zip(['rgb', 'disp', 'acc', 'weights', 'depth'], r):
                out[f'{name}_{S}_{int(white)}'] = val.numpy()
    # sample_pdf
    n = 40
    bins = (.5 * (O.coarse_z_vals(2., 6., 64, n)[:, 1:] + O.coarse_z_vals(2., 6., 64, n)[:, :-1])).clone()
    w = torch.rand(n, 62, generator=g)
    w[0] = 0.  # uniform after +1e-5
    w[1] = 0.
    w[1, 17] = 1.  # delta: many duplicate cdf values -> denom<1e-5 branch
    w[2] = 0.
    w[2, 0] = 5.  # all mass in first bin
    w[3] = 0.
    w[3, 61] = 5.  # all mass in last bin: u=1.0 edge, inds=63 -> above clamps to 62
    w[4] = 1e-9
    w[5, 10:50] = 0.
    w[6] = 1e6 * torch.rand(62, generator=g)
    w[7] = torch.rand(62, generator=g)**8
    out['pdf_bins'], out['pdf_weights'] = bins.numpy(), w.numpy()
    zs = RH.sample_pdf(bins, w, 128, det=True)
    assert torch.equal(zs, O.sample_pdf(bins, w, 128))
    out['pdf_samples'] = zs.numpy()
    out['pdf_merged'] = torch.sort(torch.cat([O.coarse_z_vals(2., 6., 64, n), zs], -1), -1)[0].numpy()
    np.savez_compressed(os.path.join(HERE, 'scan_cases.npz'), **out)
    print('scan_cases.npz', len(out), 'arrays')


if __name__ == '__main__':
    gen_r2l()
    gen_teacher()
    gen_scan_cases()
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, 'KiB')
