"""GPU: `create_data rand` shards (rays_o, rays_d, rgb) vs the CPU oracle driven by the same
restated numpy stream: ray columns bit-exact, rgb within the teacher tolerance, same shard
count / shapes / shuffle."""
import os

import numpy as np
import pytest
import torch

from oracle import r2l_oracle as O

pytestmark = pytest.mark.gpu


def oracle_create_rand(sd0, sd1, H, W, focal, n_pose, i_save, split_size, stream, render_rays=O.render_rays):
    shards, data = [], []
    for i in range(1, n_pose + 1):
        pose = stream.rand_pose()
        focal_ = focal * stream.rand_focal_scale()
        ro, rd = O.get_rays(H, W, focal_, pose[:3, :4])
        out = render_rays(sd0, sd1, ro.reshape(-1, 3).float(), rd.reshape(-1, 3).float(), white_bkgd=True)
        data.append(torch.cat([ro.reshape(-1, 3), rd.reshape(-1, 3), out['rgb_map']], -1))
        if i % i_save == 0:
            d = torch.cat(data, 0)
            ix1, ix2 = stream.permutation(d.shape[0]), stream.permutation(d.shape[0])
            d = d[ix1][ix2].numpy()
            num = d.shape[0] // split_size * split_size
            shards += [d[ix:ix + split_size] for ix in range(0, num, split_size)]
            data = []
    return shards


def test_rand_stream_matches_reference_consumption(pkg):
    from efficient_nerf_amd.create_data import RandStream
    s = RandStream()
    rs = np.random.RandomState(0)
    for _ in range(200):
        rs.rand(), rs.rand()
    theta, phi = -180 + rs.rand() * 360, -90 + rs.rand() * 90
    assert torch.equal(s.rand_pose(), O.pose_spherical(theta, phi, 4))
    assert s.rand_focal_scale() == rs.rand() + 1


def test_create_rand_shards(pkg, tmp_path):
    from efficient_nerf_amd import NeRFEngine
    from efficient_nerf_amd.create_data import RandStream, create_rand
    H = W = 12
    focal = O.focal_from_angle(W)
    sd0, sd1 = O.make_teacher_state(1), O.make_teacher_state(2)
    eng = NeRFEngine(H, W, focal).load_state_dicts(sd0, sd1)
    out = str(tmp_path / 'pseudo')
    n = create_rand(eng, H, W, focal, n_pose_kd=5, datadir_new=out, i_save=2, split_size=100, stream=RandStream())
    want = oracle_create_rand(sd0, sd1, H, W, focal, 5, 2, 100, RandStream())
    assert n == len(want) == 4  # 2 groups x (2*144 // 100) shards; the 5th pose is never flushed (as in the reference)
    for k, w in enumerate(want, 1):
        got = np.load(os.path.join(out, f'data_{k}.npy'))
        assert got.shape == (100, 9) and got.dtype == np.float32
        np.testing.assert_array_equal(got[:, :6], w[:, :6])  # rays_o, rays_d: bit-exact, same shuffle
        assert np.abs(got[:, 6:] - w[:, 6:]).max() <= 1e-4
    assert os.path.exists(os.path.join(out, 'pseudo_sample_1.png'))
    # a second run keeps the existing shards and continues the numbering (create_data.py:789-795)
    n2 = create_rand(eng, H, W, focal, n_pose_kd=2, datadir_new=out, i_save=2, split_size=100, stream=RandStream())
    assert n2 == 2 and os.path.exists(os.path.join(out, 'data_6.npy'))
    eng.close()


def _rank_create_rand(rank, world, port, out_dir, n_pose, i_save, H):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import _pkg
    _pkg.load()
    from efficient_nerf_amd import NeRFEngine, PREC_FP16_FP8, dist as D
    from efficient_nerf_amd.create_data import RandStream, create_rand
    if world > 1:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                          WORLD_SIZE=str(world), R2L_DIST_BACKEND='gloo')
        D.init()
        torch.cuda.set_device(D.local_device(rank))
    focal = O.focal_from_angle(H)
    eng = NeRFEngine(H, H, focal, precision=PREC_FP16_FP8).load_state_dicts(O.make_teacher_state(1), O.make_teacher_state(2))
    n = create_rand(eng, H, H, focal, n_pose_kd=n_pose, datadir_new=out_dir, i_save=i_save, split_size=100,
                    stream=RandStream(), log=lambda *a, **k: None)
    assert n == (n_pose // i_save) * (i_save * H * H // 100)
    eng.close()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def test_two_gpu_ranks_write_the_single_rank_directory(pkg, tmp_path):
    """config 5 sharded by pose over two processes (gloo between them, both on this GPU, the real teacher kernels):
    the shard directory is byte-identical to the one-rank run, also with i_save % world != 0 over several groups"""
    import hashlib
    import socket
    import torch.multiprocessing as mp
    H, n_pose, i_save = 12, 7, 3
    d1, d2 = str(tmp_path / 'w1'), str(tmp_path / 'w2')
    _rank_create_rand(0, 1, 0, d1, n_pose, i_save, H)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_rank_create_rand, args=(2, port, d2, n_pose, i_save, H), nprocs=2, join=True)

    def digest(d):
        return {n: hashlib.sha256(open(os.path.join(d, n), 'rb').read()).hexdigest() for n in sorted(os.listdir(d)) if n.endswith('.npy')}
    a, b = digest(d1), digest(d2)
    assert len(a) == (n_pose // i_save) * (i_save * H * H // 100) and a == b


def test_create_data_command_line(pkg, tmp_path):
    """README.md:79 of the reference: `python utils/create_data.py --create_data rand --config configs/lego.txt --teacher_ckpt X.tar
    --n_pose_kd N --datadir_kd old:new`, here `create_data.py` at the repo root with the CLI's default `--precision auto`: shards of
    the group against the oracle-driven reference stream (ray columns bit-exact, rgb within the teacher tolerance), then the reader"""
    import subprocess
    import sys
    from efficient_nerf_amd import frontend as fe
    from efficient_nerf_amd.create_data import BlenderDataset_v2, RandStream
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sd0, sd1 = O.make_teacher_state(1), O.make_teacher_state(2)
    ck = str(tmp_path / 'nerf.tar')
    fe.save_checkpoint(ck, sd0, sd1)
    out = str(tmp_path / 'pseudo')
    r = subprocess.run([sys.executable, os.path.join(root, 'create_data.py'), '--create_data', 'rand', '--config', 'configs/lego.txt',
                        '--teacher_ckpt', ck, '--n_pose_kd', '5', '--datadir_kd', f'unused:{out}', '--create_data_chunk', '2',
                        '--split_size', '100', '--H', '24', '--synthetic_poses', '1'], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert 'wrote 4 shard(s) of 100 rays; 5 poses in' in r.stdout, r.stdout[-800:]
    H = 12          # half_res of --H 24 (configs/lego.txt)
    want = oracle_create_rand(sd0, sd1, H, H, O.focal_from_angle(24) / 2., 5, 2, 100, RandStream())
    assert len(want) == 4
    for k, w in enumerate(want, 1):
        got = np.load(os.path.join(out, f'data_{k}.npy'))
        np.testing.assert_array_equal(got[:, :6], w[:, :6])
        assert np.abs(got[:, 6:] - w[:, 6:]).max() <= 1e-4
    ds = BlenderDataset_v2(out, pseudo_ratio=-1)
    assert len(ds) == 4 and all(t.shape == (100, 3) for t in ds[0])


def test_create_data_with_a_teacher_outside_the_fused_kernels(pkg, tmp_path):
    """the same command line with a 4 x 64 teacher and a 6 x 96 fine network (--netdepth / --netwidth(_fine)): the generic fp32
    layer path renders the poses, the writer and the numpy stream are the same"""
    import subprocess
    import sys
    from efficient_nerf_amd import frontend as fe
    from efficient_nerf_amd.create_data import RandStream
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sd0, sd1 = O.make_nerf_state(5, 4, 64), O.make_nerf_state(6, 6, 96)
    ck = str(tmp_path / 'nerf.tar')
    fe.save_checkpoint(ck, sd0, sd1)
    out = str(tmp_path / 'pseudo')
    r = subprocess.run([sys.executable, os.path.join(root, 'create_data.py'), '--create_data', 'rand', '--config', 'configs/lego.txt',
                        '--netdepth', '4', '--netwidth', '64', '--netdepth_fine', '6', '--netwidth_fine', '96',
                        '--teacher_ckpt', ck, '--n_pose_kd', '4', '--datadir_kd', f'unused:{out}', '--create_data_chunk', '2',
                        '--split_size', '64', '--H', '20', '--synthetic_poses', '1'], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert 'generic fp32 layer path' in r.stdout and 'wrote 6 shard(s) of 64 rays; 4 poses in' in r.stdout, r.stdout[-800:]
    H = 10
    want = oracle_create_rand(sd0, sd1, H, H, O.focal_from_angle(20) / 2., 4, 2, 64, RandStream(),
                              render_rays=lambda a, b, ro, rd, white_bkgd: O.render_rays_generic(a, b, ro, rd, white_bkgd=white_bkgd))
    assert len(want) == 6
    for k, w in enumerate(want, 1):
        got = np.load(os.path.join(out, f'data_{k}.npy'))
        np.testing.assert_array_equal(got[:, :6], w[:, :6])
        assert np.abs(got[:, 6:] - w[:, 6:]).max() <= 1e-4


def test_create_data_command_line_starts_its_own_ranks(pkg, tmp_path):
    """`python create_data.py --gpus 2 …` without torchrun: the entry script starts the two ranks itself
    (launch.py; gloo between them, both on this GPU) and the directory is byte-identical to the one-process command's"""
    import hashlib
    import subprocess
    import sys
    from efficient_nerf_amd import frontend as fe
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ck = str(tmp_path / 'nerf.tar')
    fe.save_checkpoint(ck, O.make_teacher_state(1), O.make_teacher_state(2))
    env = dict(os.environ, R2L_DIST_BACKEND='gloo')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    dirs = []
    for gpus in (1, 2):
        out = str(tmp_path / f'pseudo{gpus}')
        r = subprocess.run([sys.executable, os.path.join(root, 'create_data.py'), '--create_data', 'rand', '--config', 'configs/lego.txt',
                            '--teacher_ckpt', ck, '--n_pose_kd', '6', '--datadir_kd', f'unused:{out}', '--create_data_chunk', '3',
                            '--split_size', '100', '--H', '24', '--synthetic_poses', '1', '--gpus', str(gpus), '--launch_timeout', '500'],
                           cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        assert f'on {gpus} GPU(s)' in r.stdout, r.stdout[-800:]
        dirs.append({n: hashlib.sha256(open(os.path.join(out, n), 'rb').read()).hexdigest() for n in sorted(os.listdir(out)) if n.endswith('.npy')})
    assert len(dirs[0]) == 2 * (3 * 144 // 100) and dirs[0] == dirs[1]
