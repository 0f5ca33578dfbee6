"""CPU / gloo: the multi-rank logic of `create_data rand` (efficient-nerf_amd/create_data.py) -- pose ownership by
index inside a save group, one all-gather per group, shards written by rank k % world -- must produce a directory
that is byte-identical for every world size, also when i_save % world != 0 and over several groups (the round-1
code corrupted that case).  The teacher is replaced by a deterministic stand-in (the kernels have their own GPU
test, tests/test_create_data_gpu.py); `get_rays` is the CPU oracle's.  Also: the BlenderDataset_v2 reader."""
import hashlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class FakeTeacher:
    device = torch.device('cpu')

    def render_rays(self, rays_o, rays_d):
        m = torch.tensor([[.3, -.2, .5], [.1, .7, -.4], [-.6, .2, .3]])
        return {'rgb_map': torch.sigmoid(rays_d @ m + rays_o @ m.T * .1)}


def _run(rank, world, port, out_dir, n_pose, i_save, H, W):
    sys.path.insert(0, ROOT)
    import _pkg
    _pkg.load()
    from efficient_nerf_amd import dist as D
    from efficient_nerf_amd.create_data import RandStream, create_rand
    from oracle import r2l_oracle as O
    if world > 1:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                          WORLD_SIZE=str(world))
        D.init(backend='gloo')

    def get_rays_fn(H, W, focal, c2w, device=None):
        return O.get_rays(H, W, focal, c2w)
    n = create_rand(FakeTeacher(), H, W, O.focal_from_angle(W), n_pose, out_dir, i_save=i_save, split_size=64,
                    stream=RandStream(), log=lambda *a, **k: None, get_rays_fn=get_rays_fn)
    assert n == (n_pose // i_save) * (i_save * H * W // 64)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def _digest(d):
    out = {}
    for name in sorted(os.listdir(d)):
        if name.endswith('.npy'):
            out[name] = hashlib.sha256(open(os.path.join(d, name), 'rb').read()).hexdigest()
    return out


# (8, 8, 17): the target machine's world size -- 8 poses per save group (one per rank), two groups + a remainder pose;
# (8, 5, 11): fewer poses per group than ranks
@pytest.mark.parametrize('world,i_save,n_pose', [(2, 5, 11), (3, 5, 11), (3, 4, 9), (8, 8, 17), (8, 5, 11)])
def test_multi_rank_directory_is_identical_to_single_rank(tmp_path, world, i_save, n_pose):
    H, W = 6, 8
    d1, dn = str(tmp_path / 'w1'), str(tmp_path / f'w{world}')
    _run(0, 1, 0, d1, n_pose, i_save, H, W)
    mp.spawn(_run, args=(world, _free_port(), dn, n_pose, i_save, H, W), nprocs=world, join=True)
    a, b = _digest(d1), _digest(dn)
    assert len(a) == (n_pose // i_save) * (i_save * H * W // 64) and len(a) >= 2 * (i_save * H * W // 64)
    assert a == b
    # no all-zero rows (a slot read back without having been written shows up as zeros)
    for name in a:
        arr = np.load(os.path.join(dn, name))
        assert arr.shape == (64, 9) and arr.dtype == np.float32
        assert (np.abs(arr).sum(1) > 0).all()


def test_shard_reader_round_trip(tmp_path):
    sys.path.insert(0, ROOT)
    import _pkg
    _pkg.load()
    from efficient_nerf_amd.create_data import BlenderDataset_v2
    d = str(tmp_path / 'shards')
    _run(0, 1, 0, d, 4, 2, 6, 8)
    ds = BlenderDataset_v2(d, pseudo_ratio=-1)
    files = sorted(x for x in os.listdir(d) if x.endswith('.npy'))
    assert len(ds) == len(files) == 2 * (2 * 48 // 64)
    seen = []
    for k in range(len(ds)):
        ro, rd, rgb = ds[k]
        raw = np.load(ds.all_splits[k])
        assert ro.shape == rd.shape == rgb.shape == (64, 3)
        assert np.array_equal(torch.cat([ro, rd, rgb], -1).numpy(), raw)
        seen.append(os.path.basename(ds.all_splits[k]))
    assert sorted(seen) == files
    # original (train_*) + pseudo mixing as in the reference: num_pseudo = int(n_orig / (1 - ratio)) - n_orig
    np.save(os.path.join(d, 'train_0.npy'), np.zeros((64, 9), np.float32))
    np.random.seed(0)
    ds2 = BlenderDataset_v2(d, pseudo_ratio=0.5)
    assert len(ds2) == 2 and sum(os.path.basename(p).startswith('train_') for p in ds2.all_splits) == 1


def _load_pkg():
    sys.path.insert(0, ROOT)
    import _pkg
    _pkg.load()


@pytest.mark.parametrize('n', [0, 1, 2, 3, 17, 63, 64, 65, 129, 1000, 65536, 65537, 1 << 20])
def test_library_permutation_is_numpys_legacy_permutation(n):
    """RandStream.permutation = r2l_np_legacy_permutation (csrc/np_shuffle.hip) on the stream's own MT19937 state: the same
    indices as np.random.RandomState.permutation AND the same state afterwards (the next rand() / permutation agree), from a
    state in the middle of a block of 624 words"""
    _load_pkg()
    from efficient_nerf_amd.create_data import RandStream
    ours, ref = RandStream(seed=3, n_loader_poses=7), np.random.RandomState(3)
    for _ in range(7):
        ref.rand(), ref.rand()
    for _ in range(2):
        a, b = ours.permutation(n), ref.permutation(n)
        assert a.dtype == np.int32 and np.array_equal(a, b)
        assert ours.rs.rand() == ref.rand()
    assert torch.equal(ours.rand_pose(), RandStreamRef(ref).rand_pose())


class RandStreamRef:
    def __init__(self, rs):
        self.rs = rs

    def rand_pose(self):
        from oracle import r2l_oracle as O
        theta = -180 + self.rs.rand() * 360
        phi = -90 + self.rs.rand() * 90
        return O.pose_spherical(theta, phi, 4)


def test_permutation_rejects_bad_arguments():
    _load_pkg()
    import ctypes as C
    from efficient_nerf_amd import _lib
    L = _lib.lib()
    key = (C.c_uint * 624)()
    out = (C.c_int * 4)()
    pos = C.c_int(625)
    assert L.r2l_np_legacy_permutation(key, C.byref(pos), 4, out) != 0 and b'position' in L.r2l_last_error()
    pos = C.c_int(0)
    assert L.r2l_np_legacy_permutation(key, C.byref(pos), -1, out) != 0
    assert L.r2l_np_legacy_permutation(None, C.byref(pos), 4, out) != 0 and b'NULL' in L.r2l_last_error()


def test_shards_are_the_reference_loops_bytes(tmp_path):
    """The directory against a literal restatement of the reference's loop (utils/create_data.py:812-872) with numpy's OWN
    permutation and np.save: pose + focal draws, cat, data[ix1][ix2], split_size slices, remainder dropped, the poses behind
    the last full group never flushed -- every file byte for byte."""
    _load_pkg()
    import io
    from efficient_nerf_amd.create_data import RandStream
    from oracle import r2l_oracle as O
    H, W, n_pose, i_save, split_size = 6, 8, 7, 3, 64
    d = str(tmp_path / 'ours')
    _run(0, 1, 0, d, n_pose, i_save, H, W)
    rs = RandStream()          # only its pose / focal draws are used below; the permutations are numpy's
    teacher, focal = FakeTeacher(), O.focal_from_angle(W)
    data, split = [], 0
    for i in range(1, n_pose + 1):
        pose = rs.rand_pose()
        focal_ = focal * rs.rand_focal_scale()
        ro, rd = O.get_rays(H, W, focal_, pose[:3, :4])
        rgb = teacher.render_rays(ro.reshape(-1, 3), rd.reshape(-1, 3))['rgb_map']
        data.append(torch.cat([ro.reshape(-1, 3), rd.reshape(-1, 3), rgb], -1))
        if i % i_save == 0:
            dd = torch.cat(data, 0)
            ix1, ix2 = rs.rs.permutation(dd.shape[0]), rs.rs.permutation(dd.shape[0])
            dd = dd[ix1][ix2].numpy()
            for ix in range(0, dd.shape[0] // split_size * split_size, split_size):
                split += 1
                buf = io.BytesIO()
                np.save(buf, dd[ix:ix + split_size])
                assert open(os.path.join(d, f'data_{split}.npy'), 'rb').read() == buf.getvalue(), split
            data = []
    assert split == len([x for x in os.listdir(d) if x.endswith('.npy')]) == 2 * (3 * 48 // 64)
    assert os.path.exists(os.path.join(d, 'pseudo_sample_5.png'))


def _run_failing(rank, world, port, out_dir, result_dir):
    """rank body of the write-failure test: records what create_rand raised"""
    sys.path.insert(0, ROOT)
    import _pkg
    _pkg.load()
    from efficient_nerf_amd import dist as D
    from efficient_nerf_amd.create_data import RandStream, ShardWriteError, create_rand
    from oracle import r2l_oracle as O
    if world > 1:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                          WORLD_SIZE=str(world))
        D.init(backend='gloo')

    def get_rays_fn(H, W, focal, c2w, device=None):
        return O.get_rays(H, W, focal, c2w)
    what = 'returned'
    try:
        create_rand(FakeTeacher(), 6, 8, O.focal_from_angle(8), 40, out_dir, i_save=4, split_size=64, stream=RandStream(),
                    log=lambda *a, **k: None, get_rays_fn=get_rays_fn, writer_threads=2)
    except ShardWriteError as e:
        what = 'ShardWriteError: %s | cause %r' % (e, e.__cause__)
    open(os.path.join(result_dir, f'rank{rank}'), 'w').write(what)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [1, 2])
@pytest.mark.timeout(120)
def test_a_failed_shard_write_raises_on_every_rank_instead_of_hanging(tmp_path, world):
    """an exception inside the writer threads (ENOSPC, EIO; here: the path of shard 5 is a directory) used to skip the
    release of the host buffer -- two groups later the render loop waited for a buffer forever and, multi-rank, the peers inside
    the all-to-all.  Now the job raises ShardWriteError, on every rank, within a group or two of the failure."""
    out, res = tmp_path / 'pseudo', tmp_path / 'res'
    os.makedirs(out / 'data_5.npy')          # open(..., 'wb') of shard 5 fails (it also counts as an existing shard: numbering starts at 2)
    os.makedirs(res)
    if world == 1:
        _run_failing(0, 1, 0, str(out), str(res))
    else:
        mp.spawn(_run_failing, args=(world, _free_port(), str(out), str(res)), nprocs=world, join=True)
    got = [open(res / f'rank{r}').read() for r in range(world)]
    assert all(g.startswith('ShardWriteError') for g in got), got
    assert sum('IsADirectoryError' in g for g in got) >= 1, got                  # the rank that owns shard 5 names the cause
    if world > 1:
        assert any('another rank failed' in g for g in got), got                  # ... its peer raises with it
    # the job stopped early: far fewer than the 10 groups x 3 shards were written
    assert len([n for n in os.listdir(out) if n.endswith('.npy')]) < 20
