"""CPU: the data formats either side of the render path: PNG decoding, the Blender
test-split loader (transforms json, frames[::testskip], half_res box filter, white-background
compositing) and the test-report metrics against vectors produced by the reference's own
utils/ssim_torch.py / img2mse / mse2psnr (tests/golden/make_golden_metrics.py)."""
import json
import os
import struct
import zlib

import numpy as np
import pytest
import torch


@pytest.fixture(scope='module')
def mods(pkg):
    from efficient_nerf_amd import blender, frontend, metrics
    return blender, frontend, metrics


def encode_png(img, filters):
    """Test-side encoder with a forced filter type per row (filters[y % len])."""
    h, w, c = img.shape
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[c]
    rows = []
    prev = np.zeros(w * c, dtype=np.int32)
    for y in range(h):
        cur = img[y].reshape(-1).astype(np.int32)
        ft = filters[y % len(filters)]
        left = np.concatenate([np.zeros(c, np.int32), cur[:-c]])
        upleft = np.concatenate([np.zeros(c, np.int32), prev[:-c]])
        if ft == 0:
            pred = 0
        elif ft == 1:
            pred = left
        elif ft == 2:
            pred = prev
        elif ft == 3:
            pred = (left + prev) >> 1
        else:
            p = left + prev - upleft
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
        rows.append(bytes([ft]) + ((cur - pred) & 255).astype(np.uint8).tobytes())
        prev = cur

    def chunk(tag, data):
        return struct.pack('>I', len(data)) + tag + data + struct.pack('>I', zlib.crc32(tag + data) & 0xffffffff)

    blob = zlib.compress(b''.join(rows), 6)
    half = len(blob) // 2  # two IDAT chunks: the stream may be split anywhere
    return (b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, ctype, 0, 0, 0)) +
            chunk(b'tEXt', b'Comment\x00x') + chunk(b'IDAT', blob[:half]) + chunk(b'IDAT', blob[half:]) + chunk(b'IEND', b''))


@pytest.mark.parametrize('channels', [1, 2, 3, 4])
def test_png_decoder_all_filters(mods, tmp_path, channels):
    blender = mods[0]
    rng = np.random.default_rng(channels)
    img = rng.integers(0, 256, size=(13, 9, channels), dtype=np.uint8)
    img[3:6] = 255
    img[7] = 0
    for filters in ([0], [1], [2], [3], [4], [4, 3, 2, 1, 0]):
        p = tmp_path / f'f{channels}_{"".join(map(str, filters))}.png'
        p.write_bytes(encode_png(img, filters))
        got = blender.read_png(str(p))
        assert got.dtype == np.uint8 and np.array_equal(got, img), filters
        try:  # an independent decoder, when the box has one, must agree with the test's encoder
            from PIL import Image
            ref = np.asarray(Image.open(str(p)))
            assert np.array_equal(ref.reshape(img.shape), img)
        except ImportError:
            pass


def test_png_writer_reader_roundtrip_and_errors(mods, tmp_path):
    blender, fe, _ = mods
    rng = np.random.default_rng(0)
    for ch in (3, 4):
        img = rng.integers(0, 256, size=(20, 31, ch), dtype=np.uint8)
        fe.write_png(str(tmp_path / 'w.png'), img)
        assert np.array_equal(blender.read_png(str(tmp_path / 'w.png')), img)
    (tmp_path / 'bad.png').write_bytes(b'not a png at all')
    with pytest.raises(ValueError):
        blender.read_png(str(tmp_path / 'bad.png'))
    sixteen = (b'\x89PNG\r\n\x1a\n' + struct.pack('>I', 13) + b'IHDR' + struct.pack('>IIBBBBB', 2, 2, 16, 2, 0, 0, 0) +
               b'\x00' * 4)
    (tmp_path / 's.png').write_bytes(sixteen)
    with pytest.raises(ValueError):
        blender.read_png(str(tmp_path / 's.png'))


def make_scene(fe, root, n=6, size=16, rgba=True):
    rng = np.random.default_rng(5)
    os.makedirs(os.path.join(root, 'test'), exist_ok=True)
    imgs, frames = [], []
    for i in range(n):
        img = rng.integers(0, 256, size=(size, size, 4 if rgba else 3), dtype=np.uint8)
        fe.write_png(os.path.join(root, 'test', f'r_{i}.png'), img)
        imgs.append(img)
        pose = np.eye(4)
        pose[:3, 3] = [i, 2 * i, 4.0]
        frames.append({'file_path': f'./test/r_{i}', 'rotation': 0.1, 'transform_matrix': pose.tolist()})
    with open(os.path.join(root, 'transforms_test.json'), 'w') as fp:
        json.dump({'camera_angle_x': 0.6911112070083618, 'frames': frames}, fp)
    return np.stack(imgs)


def test_blender_loader_rules(mods, tmp_path):
    blender, fe, _ = mods
    raw = make_scene(fe, str(tmp_path))
    imgs, poses, (H, W, focal), i_split = blender.load_blender_data(str(tmp_path), half_res=False, testskip=2,
                                                                   splits=('test',))
    assert imgs.shape == (3, 16, 16, 4) and imgs.dtype == torch.float32 and poses.shape == (3, 4, 4)
    assert np.array_equal(imgs.numpy(), (raw[::2] / 255.).astype(np.float32))
    assert poses[:, 0, 3].tolist() == [0., 2., 4.] and len(i_split) == 1 and i_split[0].tolist() == [0, 1, 2]
    assert (H, W) == (16, 16) and abs(focal - .5 * 16 / np.tan(.5 * 0.6911112070083618)) < 1e-12
    # testskip == 0 means every frame (load_blender.py:52-55)
    assert blender.load_blender_data(str(tmp_path), False, 0, ('test',))[0].shape[0] == 6
    # half_res: H, W, focal halve; every output pixel is the mean of its 2x2 block
    imh, _, (H2, W2, f2), _ = blender.load_blender_data(str(tmp_path), True, 1, ('test',))
    assert (H2, W2) == (8, 8) and f2 == focal / 2. and imh.shape == (6, 8, 8, 4)
    full = (raw / 255.).astype(np.float32).astype(np.float64)
    want = full.reshape(6, 8, 2, 8, 2, 4).mean((2, 4))
    assert np.abs(imh.numpy() - want).max() <= 1e-7
    try:  # PIL's box filter on the 8-bit colour planes is the same area average (to its 8-bit, two-pass rounding;
        # on RGBA it would premultiply alpha, which neither cv2 nor the reference does)
        from PIL import Image
        box = np.asarray(Image.fromarray(raw[0][..., :3].copy()).resize((8, 8), Image.BOX)).astype(np.float64) / 255.
        assert np.abs(box - want[0][..., :3]).max() <= 1.0 / 255 + 1e-9
    except ImportError:
        pass
    # compositing (main.py:933-937)
    rgb_w = blender.composite(imgs, True)
    a = imgs[..., 3:]
    assert torch.equal(rgb_w, imgs[..., :3] * a + (1. - a)) and torch.equal(blender.composite(imgs, False), imgs[..., :3])
    rgb_only = torch.rand(2, 4, 4, 3)
    assert torch.equal(blender.composite(rgb_only, True), rgb_only)  # DONERF frames have no alpha


def test_metrics_match_reference_golden(mods, golden_dir):
    metrics = mods[2]
    g = np.load(os.path.join(golden_dir, 'metrics.npz'))
    for i in range(5):
        a, b = torch.from_numpy(g[f'a_{i}']), torch.from_numpy(g[f'b_{i}'])
        s = metrics.ssim_hwc(a, b)
        assert abs(float(s) - float(g[f'ssim_{i}'])) <= 1e-6, i
        mse = metrics.img2mse(a, b)
        assert abs(float(mse) - float(g[f'mse_{i}'])) <= 1e-9
        if float(mse) > 0:
            assert abs(float(metrics.mse2psnr(mse)) - float(g[f'psnr_{i}'][0])) <= 1e-4
    # size_average=False returns one value per image
    a = torch.from_numpy(g['a_1']).permute(2, 0, 1)[None].repeat(2, 1, 1, 1)
    b = torch.from_numpy(g['b_1']).permute(2, 0, 1)[None].repeat(2, 1, 1, 1)
    per = metrics.ssim(a, b, size_average=False)
    assert per.shape == (2,) and abs(float(per[0]) - float(g['ssim_1'])) <= 1e-6


def test_load_test_set_fallbacks(mods, tmp_path):
    _, fe, _ = mods
    a = fe.parse_args(['--dataset_type', 'blender', '--datadir', str(tmp_path / 'nowhere'), '--render_test'])
    poses, hwf, gt = fe.load_test_set(a)
    assert gt is None and poses.shape[1:] == (4, 4)  # no data mounted: synthetic poses, no GT
    make_scene(fe, str(tmp_path / 'scene'), n=4, size=8)
    b = fe.parse_args(['--dataset_type', 'blender', '--datadir', str(tmp_path / 'scene'), '--render_test',
                       '--testskip', '2', '--white_bkgd'])
    poses, hwf, gt = fe.load_test_set(b)
    assert poses.shape == (2, 4, 4) and hwf[:2] == (8, 8) and gt.shape == (2, 8, 8, 3)
    assert float(gt.min()) >= 0. and float(gt.max()) <= 1.
