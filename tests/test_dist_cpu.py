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


# -- world 8: the only size the target machine has -----------------------------------------------
class _FakeEngine:
    """stand-in for R2LEngine on the CPU: the fields and calls dist.agree_act_exponents / agree_precision use"""
    device = torch.device('cpu')

    def __init__(self, precision, exps, split=None, n_block=3):
        self.precision, self._ex, self.split_block, self.n_block = precision, list(exps), split, n_block

    def act_exponents(self):
        return list(self._ex)

    def set_act_exponents(self, ex):
        self._ex = list(ex)

    def set_precision(self, p):
        self.precision = int(p)

    def set_split_block(self, s):
        self.split_block = int(s)


def _worker8(rank, world, port, H, W, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    import _pkg
    _pkg.load()
    from efficient_nerf_amd import dist as D
    from efficient_nerf_amd import PRECISIONS
    torch.set_num_threads(1)
    D.init(backend='gloo')
    r0, r1 = D.row_shard(H, rank, world)
    # two frames of rows whose content names (frame, row, column, channel): any off-by-one in the assembly shows
    rows = torch.arange(r0, r1, dtype=torch.float32)[None, :, None, None]
    local = (torch.arange(2, dtype=torch.float32)[:, None, None, None] * 1e6 + rows * 1e3 +
             torch.arange(W, dtype=torch.float32)[None, None, :, None] * 4 + torch.arange(3, dtype=torch.float32)).reshape(2, (r1 - r0) * W, 3)
    frames = D.gather_rows(local.contiguous(), H, W, world)
    want = (torch.arange(2, dtype=torch.float32)[:, None, None, None] * 1e6 + torch.arange(H, dtype=torch.float32)[None, :, None, None] * 1e3 +
            torch.arange(W, dtype=torch.float32)[None, None, :, None] * 4 + torch.arange(3, dtype=torch.float32)).reshape(2, H * W, 3)
    assert torch.equal(frames, want), rank
    res = {}
    # per-rank exponents (each rank calibrated on its own rows) -> the element-wise maximum on every rank, in every mode with scales
    for name in ('fp16_fp8', 'fp16_e4m3', 'fp16_split', 'fp16_split8'):
        eng = _FakeEngine(PRECISIONS[name], [(rank * 3 + i) % 7 - 2 for i in range(7)], split=1)
        got = D.agree_act_exponents(eng)
        res[name] = (got, eng.act_exponents())
    eng = _FakeEngine(PRECISIONS['fp16x3_asm'], [rank] * 7)
    assert D.agree_act_exponents(eng) is None and eng.act_exponents() == [rank] * 7      # no scales: untouched
    # per-rank rungs (a noisy bisection) -> rank 0's on every rank
    eng = _FakeEngine(PRECISIONS['fp16_split'] if rank % 2 == 0 else PRECISIONS['fp16_split8'], [0] * 7, split=2 + rank % 3)
    res['agree'] = (D.agree_precision(eng), eng.precision, eng.split_block)
    eng = _FakeEngine(PRECISIONS['fp16x3_asm'] if rank == 0 else PRECISIONS['fp16_split'], [0] * 7, split=None if rank == 0 else 1)
    res['agree_x3'] = (D.agree_precision(eng), eng.precision, eng.split_block)
    torch.save(res, os.path.join(out_dir, f'res{rank}.pt'))
    D.barrier_sync()
    dist.destroy_process_group()


def test_world8_gather_and_agreement(tmp_path):
    """eight gloo ranks: 800 rows -> 100 per rank (config 4's shard) with a tiny W through gather_rows; agree_act_exponents in all four
    scaled modes (the two-part modes were skipped); agree_precision"""
    world, H, W = 8, 800, 3
    mp.spawn(_worker8, args=(world, _free_port(), H, W, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    import _pkg
    _pkg.load()
    from efficient_nerf_amd import PRECISIONS
    res = [torch.load(os.path.join(str(tmp_path), f'res{r}.pt')) for r in range(world)]
    want = [max((r * 3 + i) % 7 - 2 for r in range(world)) for i in range(7)]
    for name in ('fp16_fp8', 'fp16_e4m3', 'fp16_split', 'fp16_split8'):
        for r in range(world):
            assert res[r][name] == (want, want), (name, r, res[r][name])
    for r in range(world):
        assert res[r]['agree'] == ((PRECISIONS['fp16_split'], 2), PRECISIONS['fp16_split'], 2), res[r]['agree']
        assert res[r]['agree_x3'] == ((PRECISIONS['fp16x3_asm'], -1), PRECISIONS['fp16x3_asm'], None), res[r]['agree_x3']


def test_agree_act_exponents_explicit_split_world2(tmp_path):
    """`--precision fp16_split` on two ranks ends with identical exponents on both"""
    world = 2
    mp.spawn(_worker8, args=(world, _free_port(), 8, 3, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(str(tmp_path), f'res{r}.pt')) for r in range(world)]
    assert res[0]['fp16_split'] == res[1]['fp16_split'] and res[0]['fp16_split'][0] is not None
