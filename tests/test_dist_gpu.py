"""GPU: the library's collective (r2l_gather_image over RCCL) executes on hardware.  One rank on one GPU is what a
1-GPU box allows: ncclCommInitRank(world = 1) + the grouped ncclAllGather launch, frame-major output; the N > 1 data
movement is covered by the gloo tests (tests/test_dist_cpu.py) and runs on the driver's 8-GPU node."""
import numpy as np
import pytest
import torch

from oracle import r2l_oracle as O

pytestmark = pytest.mark.gpu


def test_rccl_gather_one_rank(pkg):
    from efficient_nerf_amd import dist as D
    F, H, W = 3, 10, 7
    g = torch.Generator().manual_seed(0)
    local = torch.rand(F, H * W, 3, generator=g).cuda()
    out = D.gather_rows(local, H, W, 1, force_collective=True)
    torch.cuda.synchronize()
    assert out.data_ptr() != local.data_ptr() and torch.equal(out, local)
    # the communicator and the output buffer are reused
    local2 = torch.rand(F, H * W, 3, generator=g).cuda()
    out2 = D.gather_rows(local2, H, W, 1, force_collective=True)
    torch.cuda.synchronize()
    assert out2.data_ptr() == out.data_ptr() and torch.equal(out2, local2)


def test_rendered_rows_through_the_collective(pkg):
    """a frame rendered as two row ranges, each pushed through the 1-rank collective, equals the full frame"""
    from efficient_nerf_amd import R2LEngine, dist as D
    H = W = 16
    sd = O.make_r2l_state(seed=0, netdepth=8)
    eng = R2LEngine(H, W, O.focal_from_angle(W), n_block=3).load_state_dict(sd)
    c2w = O.pose_spherical(10., -30., 4.)
    full = eng.render(c2w)
    got = D.gather_rows(full[None], H, W, 1, force_collective=True)[0]
    assert torch.equal(got, full)
    eng.close()
