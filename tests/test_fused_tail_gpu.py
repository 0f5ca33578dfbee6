"""The body kernel's fused tail (csrc/gen/body_gen.py fused_tail_text: rgb = sigmoid(W_t (x + h) + b_t) straight from the
residual stream, model/nerf_raybased.py:539-544) against the three-launch form it replaces (x image to HBM,
r2l_tail_kernel) on the same context, against the CPU oracle, and at the edges of the output buffer."""
import pytest
import torch

from oracle import r2l_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('H,nb', [(40, 3), (96, 43)])
def test_fused_tail_matches_three_launch_form(pkg, H, nb):
    from efficient_nerf_amd import PREC_FP16_FP8, R2LEngine
    focal = O.focal_from_angle(H)
    sd = O.make_r2l_state(seed=5, netdepth=2 + 2 * nb)
    c2w = O.pose_spherical(25., -35., 4.)
    eng = R2LEngine(H, H, focal, n_block=nb, precision=PREC_FP16_FP8).load_state_dict(sd)
    fused = eng.render(c2w).cpu()                     # also fixes the activation exponents for both forms
    split = eng._set_fused_tail(0).render(c2w).cpu()
    again = eng._set_fused_tail(1).render(c2w).cpu()
    assert torch.equal(again, fused)
    # same fp32 sums in another order, v_exp_f32 / v_rcp_f32 instead of expf and a division
    assert (fused - split).abs().max().item() <= 5e-7
    ref = O.r2l_render(sd, H, H, focal, c2w)
    assert (fused - ref).abs().max().item() <= 1e-4
    assert (split - ref).abs().max().item() <= 1e-4
    eng.close()


def test_fused_tail_writes_only_its_rays(pkg):
    """ragged ray counts: the 12-byte rows behind the call's last ray stay untouched, rows of several poses are dense"""
    from efficient_nerf_amd import PREC_FP16_FP8, R2LEngine
    H, nb = 40, 4
    focal = O.focal_from_angle(H)
    sd = O.make_r2l_state(seed=9, netdepth=2 + 2 * nb)
    c2w = O.pose_spherical(-100., -20., 4.)
    ro, rd = O.get_rays(H, H, focal, c2w)
    ro, rd = ro.reshape(-1, 3).float().contiguous().cuda(), rd.reshape(-1, 3).float().contiguous().cuda()
    eng = R2LEngine(H, H, focal, n_block=nb, precision=PREC_FP16_FP8).load_state_dict(sd)
    full = eng.render_rays(ro, rd).cpu()
    for n in (1, 33, 127, 130, 1000):
        buf = torch.full((n + 200, 3), -7.0, device='cuda')
        eng.render_rays(ro[:n].contiguous(), rd[:n].contiguous(), out=buf[:n])
        got = buf.cpu()
        assert torch.equal(got[:n], full[:n]), n
        assert (got[n:] == -7.0).all(), n
    # two poses, a row range that is not a multiple of the tile: [P, rows*W, 3] dense
    poses = torch.stack([torch.as_tensor(O.pose_spherical(t, -30., 4.))[:3, :4].float() for t in (10., 200.)]).cuda()
    out = eng.render_batch(poses, rows=(3, 20)).cpu()
    for i, t in enumerate((10., 200.)):
        ref = O.r2l_render(sd, H, H, focal, O.pose_spherical(t, -30., 4.), rows=(3, 20))
        assert (out[i] - ref).abs().max().item() <= 1e-4
    eng.close()


def test_call_larger_than_one_launch_slice(pkg):
    """fp16_fp8 renders a call in slices of 8,192 ray tiles (csrc/r2l_capi.hip launch_split): the second slice's tiles
    must land behind the first slice's rows, and a ray's result must not depend on the slice it falls in"""
    from efficient_nerf_amd import PREC_FP16_FP8, R2LEngine
    H, nb = 64, 1
    focal = O.focal_from_angle(H)
    sd = O.make_r2l_state(seed=17, netdepth=2 + 2 * nb)
    c2w = O.pose_spherical(140., -25., 4.)
    ro, rd = O.get_rays(H, H, focal, c2w)
    ro, rd = ro.reshape(-1, 3).float().contiguous().cuda(), rd.reshape(-1, 3).float().contiguous().cuda()
    n = 8192 * 128 + 300                                     # one full slice + 2.3 tiles
    idx = torch.arange(n, device='cuda') % ro.shape[0]
    big_o, big_d = ro[idx].contiguous(), rd[idx].contiguous()
    eng = R2LEngine(H, H, focal, n_block=nb, precision=PREC_FP16_FP8).load_state_dict(sd)
    frame = eng.render_rays(ro, rd)                          # 4,096 rays; fixes the exponents
    buf = torch.full((n + 64, 3), -7.0, device='cuda')
    eng.render_rays(big_o, big_d, out=buf[:n])
    assert (buf[n:] == -7.0).all()
    assert torch.equal(buf[:n], frame[idx])                  # every copy of a ray, in either slice, bit for bit
    ref = O.r2l_render(sd, H, H, focal, c2w)
    assert (frame.cpu() - ref).abs().max().item() <= 1e-4
    for mode in (0,):                                        # the three-launch form slices the same way
        eng._set_fused_tail(mode)
        buf.fill_(-7.0)
        eng.render_rays(big_o, big_d, out=buf[:n])
        assert (buf[n:] == -7.0).all()
        assert (buf[:n] - frame[idx]).abs().max().item() <= 5e-7
    eng.close()
