"""The library enqueues on the caller's stream and never synchronises (include/r2l_hip.h): a warmed-up render must
therefore be capturable into a HIP graph and replay with new pose / ray contents in the same device buffers."""
import pytest
import torch

from oracle import r2l_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('prec', ['fp16_fp8', 'fp16x3', 'fp16_e4m3', 'fp16x3_asm'])
def test_r2l_render_in_a_hip_graph(pkg, prec):
    from efficient_nerf_amd import PRECISIONS, R2LEngine
    H, nb = 64, 4
    focal = O.focal_from_angle(H)
    sd = O.make_r2l_state(seed=2, netdepth=2 + 2 * nb)
    eng = R2LEngine(H, H, focal, n_block=nb, precision=PRECISIONS[prec]).load_state_dict(sd)
    poses = [torch.as_tensor(O.pose_spherical(t, -30., 4.))[:3, :4].float().contiguous() for t in (0., 90., 215.)]
    static_pose = poses[0].cuda().clone().reshape(1, 3, 4)
    out = torch.empty((1, H * H, 3), device='cuda')
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                # warm-up on the capture stream: buffers allocated, exponents measured
        for _ in range(2):
            eng.render_batch(static_pose, out=out)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        eng.render_batch(static_pose, out=out)
    for p in poses:
        static_pose.copy_(p.reshape(1, 3, 4))
        g.replay()
        torch.cuda.synchronize()
        direct = eng.render_batch(p.cuda().reshape(1, 3, 4))
        assert torch.equal(out, direct)
        ref = O.r2l_render(sd, H, H, focal, p)
        assert (out[0].cpu() - ref).abs().max().item() <= 1e-4
    eng.close()


def test_teacher_mixed_rung_with_the_skipped_coarse_view_branch_in_a_hip_graph(pkg):
    """round 6: the rung trained teachers get (coarse fp16x3_asm without its view branch, fine fp16_mix) captures and replays as well: the
    stream without the view branch is packed when nerf_set_skip_rgb0 / the weights / the mode are set, not inside a render"""
    from efficient_nerf_amd import NeRFEngine, PRECISIONS
    H = 32
    eng = NeRFEngine(H, H, O.focal_from_angle(H), precision=PRECISIONS['fp16_mix']).load_state_dicts(O.make_teacher_state(1), O.make_teacher_state(2))
    eng.set_skip_rgb0(True)
    ro, rd = O.get_rays(H, H, eng.focal, O.pose_spherical(25., -40., 4.)[:3, :4])
    ro, rd = ro.reshape(-1, 3).float().contiguous().cuda(), rd.reshape(-1, 3).float().contiguous().cuda()
    want = {k: v.clone() for k, v in eng.render_rays(ro, rd).items()}           # warm-up: temporaries allocated
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        eng.render_rays(ro, rd)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        got = eng.render_rays(ro, rd)
    for k in got:
        got[k].zero_()
    g.replay()
    torch.cuda.synchronize()
    for k in ('rgb_map', 'acc_map', 'depth_map'):
        assert torch.equal(got[k], want[k]), k
    eng.close()


def test_teacher_render_rays_in_a_hip_graph(pkg):
    """coarse + fine networks, raw2outputs, sample_pdf, merge: eight launches, one graph"""
    from efficient_nerf_amd import NeRFEngine, PREC_FP16_FP8
    H = 24
    focal = O.focal_from_angle(H)
    sd1, sd2 = O.make_teacher_state(1), O.make_teacher_state(2)
    eng = NeRFEngine(H, H, focal, precision=PREC_FP16_FP8).load_state_dicts(sd1, sd2)
    rays = []
    for t in (30., 160.):
        ro, rd = O.get_rays(H, H, focal, O.pose_spherical(t, -30., 4.))
        rays.append((ro.reshape(-1, 3).float().contiguous().cuda(), rd.reshape(-1, 3).float().contiguous().cuda()))
    so, sdir = rays[0][0].clone(), rays[0][1].clone()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(2):
            eng.render_rays(so, sdir)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        ret = eng.render_rays(so, sdir)
    for ro, rd in rays[::-1]:
        so.copy_(ro)
        sdir.copy_(rd)
        g.replay()
        torch.cuda.synchronize()
        direct = eng.render_rays(ro, rd)
        for k in ('rgb_map', 'disp_map', 'acc_map', 'depth_map'):
            assert torch.equal(ret[k], direct[k]), k
    eng.close()
