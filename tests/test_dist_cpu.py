"""CPU / gloo, world_size 2: the N>1 path of bench.py and the front-end (row sharding + ONE
all-gather assembling the frames).  The renderer plugged in here is the CPU oracle (tests
may use it as the checker); on the GPU box the same `dist` code runs over RCCL."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, H, W, n_frames, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    import _pkg
    _pkg.load()
    from efficient_nerf_amd import dist as D
    from oracle import r2l_oracle as O
    torch.set_num_threads(2)
    r, lr, w = D.init(backend='gloo')
    assert (r, w) == (rank, world)
    sd = O.make_r2l_state(seed=3, netdepth=4)  # 1 ResMLP block: fast on CPU
    focal = O.focal_from_angle(W)
    poses = O.novel_poses(5)[:n_frames]
    r0, r1 = D.row_shard(H, rank, world)
    local = torch.stack([O.r2l_render(sd, H, W, focal, p, rows=(r0, r1)) for p in poses], 0)  # [F, rows*W, 3]
    frames = D.gather_rows(local, H, W, world)
    D.barrier_sync()
    torch.save(local, os.path.join(out_dir, f'local{rank}.pt'))
    if rank == 0:
        full = torch.stack([O.r2l_render(sd, H, W, focal, p) for p in poses], 0)
        torch.save({'frames': frames, 'full': full}, os.path.join(out_dir, 'out.pt'))
    # every rank holds the assembled frames
    chk = torch.tensor([frames.double().sum().item()], dtype=torch.float64)
    lst = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(lst, chk)
    assert all(torch.equal(lst[0], x) for x in lst)
    dist.destroy_process_group()


@pytest.mark.parametrize('H,W', [(8, 8), (7, 5)])  # equal shards and ragged shards (H % world != 0)
def test_row_shard_all_gather_world2(tmp_path, H, W):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), H, W, 3, str(tmp_path)), nprocs=world, join=True)
    out = torch.load(os.path.join(str(tmp_path), 'out.pt'))
    assert out['frames'].shape == (3, H * W, 3)
    # the collective moves bytes: assembled == concatenation of the shards, bit for bit
    locs = [torch.load(os.path.join(str(tmp_path), f'local{r}.pt')) for r in range(world)]
    assert torch.equal(out['frames'], torch.cat(locs, 1))
    # and equals the un-sharded render up to the CPU BLAS's batch-size dependent summation order
    assert (out['frames'] - out['full']).abs().max() < 1e-6


def test_row_shard_partition():
    sys.path.insert(0, ROOT)
    import _pkg
    _pkg.load()
    from efficient_nerf_amd import dist as D
    for H in (1, 7, 8, 400, 800):
        for world in (1, 2, 3, 4, 8):
            spans = [D.row_shard(H, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == H
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
