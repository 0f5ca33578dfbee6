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


def test_gather_into_a_frame_stack_keeps_every_frame(pkg):
    """frontend.render_path keeps N frames: the collective's own buffer is reused by the next call (a kept view of it
    would show the LAST frame N times), so each frame is gathered straight into its slot of a pre-allocated stack"""
    from efficient_nerf_amd import dist as D
    H, W, N = 6, 5, 4
    g = torch.Generator().manual_seed(1)
    frames = [torch.rand(1, H * W, 3, generator=g).cuda() for _ in range(N)]
    stack = torch.empty((N, H, W, 3), device='cuda')
    views = [D.gather_rows(f, H, W, 1, force_collective=True, out=stack[i])[0] for i, f in enumerate(frames)]
    torch.cuda.synchronize()
    for i in range(N):
        assert views[i].data_ptr() == stack[i].data_ptr()
        assert torch.equal(stack[i].reshape(-1, 3), frames[i][0])
    assert not torch.equal(stack[0], stack[N - 1])
    reused = [D.gather_rows(f, H, W, 1, force_collective=True)[0] for f in frames]      # without out=: one buffer
    assert len({r.data_ptr() for r in reused}) == 1


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


def _agree_worker(rank, world, port, out_dir):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), R2L_DIST_BACKEND='gloo')
    import _pkg
    _pkg.load()
    from efficient_nerf_amd import PREC_FP16_FP8, R2LEngine, dist as D
    D.init()
    torch.cuda.set_device(D.local_device(rank))
    H, nb = 64, 8
    focal = O.focal_from_angle(H)
    sd = O.make_r2l_state(seed=6, netdepth=2 + 2 * nb)
    for k in sd:                                   # growing activations: the two row shards see different ranges
        if k.startswith('body.') and k.endswith('weight'):
            sd[k] = sd[k] * 1.4
    c2w = O.pose_spherical(35., -60., 4.)
    eng = R2LEngine(H, H, focal, n_block=nb, precision=PREC_FP16_FP8).load_state_dict(sd)
    r0, r1 = D.row_shard(H, rank, world)
    eng.render(c2w, rows=(r0, r1))
    mine = eng.act_exponents()
    agreed = D.agree_act_exponents(eng)
    assert agreed == eng.act_exponents() and all(a >= m for a, m in zip(agreed, mine))
    local = eng.render(c2w, rows=(r0, r1))
    frame = D.gather_rows(local[None], H, H, world)[0]
    own = eng.render(c2w)
    torch.save({'mine': mine, 'agreed': agreed, 'equal': bool(torch.equal(frame, own)), 'frame': frame.cpu()},
               os.path.join(out_dir, f'r{rank}.pt'))
    D.barrier_sync()
    eng.close()
    import torch.distributed as dist
    dist.destroy_process_group()


def test_two_ranks_agree_on_activation_exponents(pkg, tmp_path):
    """two processes (gloo between them, both on this GPU) render the two row shards of a frame in fp16_fp8: after
    dist.agree_act_exponents both use the element-wise maximum of what they measured, and the assembled frame equals each
    rank's own render of all rows bit for bit (bench.py's gather_check, frontend.render_path)"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_agree_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(tmp_path / f'r{k}.pt') for k in range(2)]
    assert r[0]['agreed'] == r[1]['agreed'] == [max(a, b) for a, b in zip(r[0]['mine'], r[1]['mine'])]
    assert r[0]['equal'] and r[1]['equal'] and torch.equal(r[0]['frame'], r[1]['frame'])
    ref = O.r2l_render(O_state_14(), 64, 64, O.focal_from_angle(64), O.pose_spherical(35., -60., 4.))
    # NOT the 1e-4 contract: this network (body x 1.4) is beyond what an explicit fp16_fp8 holds to 1e-4 by design -- the test is
    # about rank agreement (asserted bit for bit above); this line only keeps a broken assembly from passing as "agreed"
    sanity_not_contract = 5e-4
    assert (r[0]['frame'] - ref).abs().max().item() <= sanity_not_contract


def O_state_14():
    sd = O.make_r2l_state(seed=6, netdepth=18)
    for k in sd:
        if k.startswith('body.') and k.endswith('weight'):
            sd[k] = sd[k] * 1.4
    return sd


def _watch_worker(rank, world, port, out_dir):
    import os
    import sys
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), R2L_DIST_BACKEND='gloo')
    import _pkg
    _pkg.load()
    from efficient_nerf_amd import PREC_FP16_FP8, PREC_FP16X3_ASM, R2LEngine, dist as D, get_rays
    from efficient_nerf_amd import frontend as fe
    D.init()
    torch.cuda.set_device(D.local_device(rank))
    z = np.load(os.path.join(root, 'tests', 'golden', 'trained_like', 'student_w256d88.npz'))
    ssd = {k: torch.from_numpy(z[k]).clone() for k in z.files}
    for k in ssd:                     # the scaled trained-like student: max|a| 3.9, 1.3e-4 off in fp16_fp8 (tests/test_split_gpu.py)
        if k.startswith('head.') or (k.startswith('body.') and k.endswith('bias')):
            ssd[k] = ssd[k] / 32.
        elif k == 'tail.0.weight':
            ssd[k] = ssd[k] * 32.
    H = 400
    focal = O.focal_from_angle(H)
    test = O.novel_poses(200)
    pose = test[0][:3, :4]
    ro, rd = (t.reshape(-1, 3) for t in get_rays(H, H, focal, pose, device='cuda'))
    eng = R2LEngine(H, H, focal, 2., 6., n_block=43, use_residual=True).load_state_dict(ssd)
    eng.set_precision(PREC_FP16_FP8)
    eng.calibrate_on(c2w=pose)
    got8 = eng.render(pose).clone()
    eng.set_precision(PREC_FP16X3_ASM)
    easy = torch.nonzero((got8 - eng.render(pose)).abs().max(-1)[0] <= 2e-5).flatten()[:65536]
    rung = eng.choose_precision(rays=(ro[easy].contiguous(), rd[easy].contiguous()))[0]
    D.agree_precision(eng)
    lines, stats = [], {}
    rgbs, _ = fe.render_path([test[i] for i in (0, 67, 133)], (H, H, focal), 'R2L', eng, log=lines.append, stats=stats)
    torch.save({'rung': rung, 'end': (eng.precision_name, eng.split_block), 'watch': stats.get('split_watch'), 'frames': rgbs.cpu(),
                'exps': eng.act_exponents(), 'log': [ln for ln in lines if 'precision' in ln]}, os.path.join(out_dir, f'w{rank}.pt'))
    D.barrier_sync()
    eng.close()
    import torch.distributed as dist
    dist.destroy_process_group()


def test_two_ranks_fall_back_together_when_the_rgb_watch_misses(pkg, tmp_path):
    """round 6: fp16_fp8 chosen by `auto` on a probe that does not see the hard rays (tests/test_split_gpu.py) under row sharding -- two
    processes (gloo, both on this GPU) render the two row shards of three frames through frontend.render_path: the rgb watch misses on
    the first batch, rank 0's measured rung is adopted by both (dist.agree_precision), both ranks end in the same mode, split and
    exponents, hold the same assembled frames, and those are inside 1e-4 of the CPU oracle"""
    import socket
    import numpy as np
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_watch_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(tmp_path / f'w{k}.pt') for k in range(2)]
    print(r[0]['log'], r[0]['end'], r[0]['watch'])
    assert r[0]['rung'] == r[1]['rung'] == 'fp16_fp8'
    assert r[0]['end'] == r[1]['end'] and r[0]['end'][0] in ('fp16_split', 'fp16_split8', 'fp16x3_asm') and r[0]['exps'] == r[1]['exps']
    assert len(r[0]['watch']['fallbacks']) >= 1 and len(r[0]['watch']['fallbacks']) == len(r[1]['watch']['fallbacks'])
    assert torch.equal(r[0]['frames'], r[1]['frames'])
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'trained_like', 'student_w256d88.npz'))
    ssd = {k: torch.from_numpy(z[k]) for k in z.files}           # the unscaled network: the same function
    H = 400
    test = O.novel_poses(200)
    for j, pi in enumerate((0, 67, 133)):
        want = O.r2l_render(ssd, H, H, O.focal_from_angle(H), test[pi][:3, :4], rows=(0, H, 16), chunk=16384)
        err = (r[0]['frames'][j][::16].reshape(-1, 3) - want).abs().max().item()
        print(f'frame {j}: {err:.2e} from the CPU oracle')
        assert err <= 1e-4
