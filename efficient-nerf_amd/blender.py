"""Blender (nerf_synthetic) test-set reader for `--render_test`: transforms_{split}.json, the
RGBA PNG frames, half-resolution box filter and white-background compositing.

Mirrors dataset/load_blender.py:31-120 and main.py:922-937 of the reference for the parts the
render path consumes (images, poses, [H, W, focal], split indices).  The reference reads PNGs
with imageio and halves them with cv2.INTER_AREA; neither is a dependency here: `read_png` is a
self-contained decoder (8-bit gray / RGB / RGBA, gray+alpha, all five filters, non-interlaced)
and `half_res_area` is the exact 2x2 mean INTER_AREA computes for a factor of two.
"""
import json
import os
import struct
import zlib

import numpy as np
import torch

_PNG_SIG = b'\x89PNG\r\n\x1a\n'
_CHANNELS = {0: 1, 2: 3, 4: 2, 6: 4}


def read_png(path):
    """Decode an 8-bit non-interlaced PNG to a uint8 array [H, W, C] (C = 1, 2, 3 or 4)."""
    with open(path, 'rb') as f:
        data = f.read()
    if data[:8] != _PNG_SIG:
        raise ValueError(f'{path}: not a PNG file')
    pos, idat, hdr = 8, [], None
    while pos + 8 <= len(data):
        n, tag = struct.unpack('>I4s', data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if tag == b'IHDR':
            hdr = struct.unpack('>IIBBBBB', body)
        elif tag == b'IDAT':
            idat.append(body)
        elif tag == b'IEND':
            break
        pos += 12 + n
    if hdr is None:
        raise ValueError(f'{path}: no IHDR chunk')
    w, h, depth, ctype, _, _, interlace = hdr
    if depth != 8 or ctype not in _CHANNELS or interlace != 0:
        raise ValueError(f'{path}: unsupported PNG (bit depth {depth}, colour type {ctype}, interlace {interlace}); '
                         'only 8-bit gray/RGB/RGBA non-interlaced files are read')
    c = _CHANNELS[ctype]
    stride = w * c
    raw = np.frombuffer(zlib.decompress(b''.join(idat)), dtype=np.uint8)
    if raw.size != h * (stride + 1):
        raise ValueError(f'{path}: truncated image data')
    rows = raw.reshape(h, stride + 1)
    out = np.zeros((h, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int32)
    for y in range(h):
        ft = int(rows[y, 0])
        line = rows[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:  # Up
            cur = (line + prev) & 255
        elif ft in (1, 3, 4):  # Sub / Average / Paeth: serial along the row, vectorised over the c channels
            cur = np.zeros(stride, dtype=np.int32)
            left = np.zeros(c, dtype=np.int32)
            upleft = np.zeros(c, dtype=np.int32)
            for x in range(0, stride, c):
                up = prev[x:x + c]
                if ft == 1:
                    pred = left
                elif ft == 3:
                    pred = (left + up) >> 1
                else:
                    p = left + up - upleft
                    pa, pb, pc = np.abs(p - left), np.abs(p - up), np.abs(p - upleft)
                    pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, upleft))
                left = (line[x:x + c] + pred) & 255
                cur[x:x + c] = left
                upleft = up
        else:
            raise ValueError(f'{path}: bad filter type {ft} in row {y}')
        out[y] = cur
        prev = cur
    return out.reshape(h, w, c)


def half_res_area(imgs):
    """cv2.resize(img, (W/2, H/2), interpolation=cv2.INTER_AREA) for an exact factor of two = the
    mean of each 2x2 block (dataset/load_blender.py:105-115).  imgs: float32 [N, H, W, C]."""
    n, h, w, c = imgs.shape
    v = imgs[:, :h // 2 * 2, :w // 2 * 2].reshape(n, h // 2, 2, w // 2, 2, c)
    return ((v[:, :, 0, :, 0] + v[:, :, 0, :, 1]) + (v[:, :, 1, :, 0] + v[:, :, 1, :, 1])) * np.float32(0.25)


def load_blender_data(basedir, half_res=False, testskip=1, splits=('train', 'val', 'test')):
    """dataset/load_blender.py:31-120: frames[::skip] per split (skip = 1 for train or when
    testskip == 0), imgs / 255 as float32 with all channels kept, poses float32 [N,4,4],
    focal = .5 * W / tan(.5 * camera_angle_x), half_res halves H, W, focal.  Returns
    (imgs [N,H,W,C] f32 tensor, poses [N,4,4] tensor, [H, W, focal], i_split list)."""
    all_imgs, all_poses, counts, meta = [], [], [0], None
    for s in splits:
        with open(os.path.join(basedir, f'transforms_{s}.json')) as fp:
            meta = json.load(fp)
        skip = 1 if (s == 'train' or testskip == 0) else testskip
        imgs, poses = [], []
        for frame in meta['frames'][::skip]:
            imgs.append(read_png(os.path.join(basedir, frame['file_path'] + '.png')))
            poses.append(np.array(frame['transform_matrix']))
        imgs = (np.array(imgs) / 255.).astype(np.float32)
        counts.append(counts[-1] + imgs.shape[0])
        all_imgs.append(imgs)
        all_poses.append(np.array(poses).astype(np.float32))
    i_split = [np.arange(counts[i], counts[i + 1]) for i in range(len(splits))]
    imgs = np.concatenate(all_imgs, 0)
    poses = np.concatenate(all_poses, 0)
    H, W = imgs[0].shape[:2]
    if 'camera_angle_x' in meta:
        angle = float(meta['camera_angle_x'])
    else:  # DONERF layout (load_blender.py:84-88)
        with open(os.path.join(basedir, 'dataset_info.json')) as fp:
            angle = float(json.load(fp)['camera_angle_x'])
    focal = .5 * W / np.tan(.5 * angle)
    if half_res:
        H, W, focal = H // 2, W // 2, focal / 2.
        imgs = half_res_area(imgs)
    return torch.from_numpy(np.ascontiguousarray(imgs)), torch.from_numpy(poses), [H, W, focal], i_split


def composite(images, white_bkgd):
    """main.py:933-937: RGBA -> RGB on white (rgb * a + (1 - a)) or plain RGB."""
    if images.shape[-1] < 4:
        return images[..., :3]
    if white_bkgd:
        return images[..., :3] * images[..., -1:] + (1. - images[..., -1:])
    return images[..., :3]
