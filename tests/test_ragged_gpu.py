"""Ragged launch sizes of the two generated kernels (r2l_body_kernel behind r2l_render_rays, nerf_chain_kernel behind
nerf_run_network): ray / point counts around the 128-wide tile (1, one short of a tile, one over, more tiles than CUs
... ) in fp16_fp8, against the same rays inside a bigger launch (bit for bit: a ray's result must not depend on its
tile mates) and against the CPU oracle (model/nerf_raybased.py:94-126, 377-401)."""
import pytest
import torch

from oracle import r2l_oracle as O

pytestmark = pytest.mark.gpu


def test_r2l_given_rays_ragged_counts(pkg):
    from efficient_nerf_amd import PREC_FP16_FP8, R2LEngine
    H, nb = 40, 6
    focal = O.focal_from_angle(H)
    sd = O.make_r2l_state(seed=13, netdepth=2 + 2 * nb)
    c2w = O.pose_spherical(70., -25., 4.)
    ro, rd = O.get_rays(H, H, focal, c2w)
    ro, rd = ro.reshape(-1, 3).float().contiguous().cuda(), rd.reshape(-1, 3).float().contiguous().cuda()
    eng = R2LEngine(H, H, focal, n_block=nb, precision=PREC_FP16_FP8).load_state_dict(sd)
    full = eng.render_rays(ro, rd).cpu()                       # 1,600 rays = 12.5 tiles; also fixes the exponents
    ref = O.r2l_render(sd, H, H, focal, c2w)
    assert (full - ref).abs().max() <= 1e-4
    for n in (1, 31, 32, 127, 128, 129, 257, 1599):
        part = eng.render_rays(ro[:n].contiguous(), rd[:n].contiguous()).cpu()
        assert part.shape == (n, 3) and torch.equal(part, full[:n]), n
    off = eng.render_rays(ro[777:1000].contiguous(), rd[777:1000].contiguous()).cpu()   # other tile mates, other lanes
    assert torch.equal(off, full[777:1000])
    eng.close()


@pytest.mark.parametrize('prec', ['fp16_fp8', 'fp16x1', 'fp16x3_asm'])
@pytest.mark.parametrize('n,S', [(1, 3), (2, 64), (1, 127), (3, 43), (5, 192), (700, 64)])
def test_teacher_chain_ragged_point_counts(pkg, n, S, prec):
    """the three generated builds of the layer chain: with bf6 terms (128-point tiles), the single pass as one statement with its ray
    loads, embedding and raw stores in the stream (256-point tiles: nerf_chain_emb_kernel's own point -> (ray, sample) division and
    masked stores), and in three passes (128-point tiles)"""
    from efficient_nerf_amd import NeRFEngine, PRECISIONS
    H = 32
    focal = O.focal_from_angle(H)
    sd = O.make_teacher_state(2)
    c2w = O.pose_spherical(-40., -15., 4.)
    ro, rd = O.get_rays(H, H, focal, c2w)
    ro, rd = ro.reshape(-1, 3).float()[:n].contiguous(), rd.reshape(-1, 3).float()[:n].contiguous()
    g = torch.Generator().manual_seed(n * 1000 + S)
    z = (2. + 4. * torch.rand(n, S, generator=g)).sort(-1).values.contiguous()
    eng = NeRFEngine(H, H, focal, precision=PRECISIONS[prec]).load_state_dicts(sd, sd)
    raw = eng.run_network(1, ro.cuda(), rd.cuda(), z.cuda()).cpu()
    assert raw.shape == (n, S, 4)
    pts = ro[:, None, :] + rd[:, None, :] * z[..., None]
    ref = O.run_network(sd, pts, rd / rd.norm(dim=-1, keepdim=True))
    err = (raw - ref).abs().max().item()
    assert err <= 2e-4, err
    if n > 1:      # the same points in a different launch shape: bit for bit
        raw1 = eng.run_network(1, ro[1:].contiguous().cuda(), rd[1:].contiguous().cuda(), z[1:].contiguous().cuda()).cpu()
        assert torch.equal(raw1, raw[1:])
    eng.close()
