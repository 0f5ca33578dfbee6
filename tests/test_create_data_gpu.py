"""GPU: `create_data rand` shards (rays_o, rays_d, rgb) vs the CPU oracle driven by the same
restated numpy stream: ray columns bit-exact, rgb within the teacher tolerance, same shard
count / shapes / shuffle."""
import os

import numpy as np
import pytest
import torch

from oracle import r2l_oracle as O

pytestmark = pytest.mark.gpu


def oracle_create_rand(sd0, sd1, H, W, focal, n_pose, i_save, split_size, stream):
    shards, data = [], []
    for i in range(1, n_pose + 1):
        pose = stream.rand_pose()
        focal_ = focal * stream.rand_focal_scale()
        ro, rd = O.get_rays(H, W, focal_, pose[:3, :4])
        out = O.render_rays(sd0, sd1, ro.reshape(-1, 3).float(), rd.reshape(-1, 3).float(), white_bkgd=True)
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
