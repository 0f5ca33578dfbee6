"""GPU: the drop-in boundary without Python in the loop -- examples/r2l_render.c (plain C, gcc,
no torch) loads weights from a flat float32 file, renders through the C-ABI and writes the RGB
floats; the frame must match the CPU oracle like the ctypes path does."""
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import r2l_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('precision,tol', [(0, 1e-4), (2, 1e-4)])
def test_c_client_renders_like_the_oracle(built_lib, tmp_path, precision, tol):
    exe = os.path.join(ROOT, 'examples', 'r2l_render')
    if not os.path.exists(exe):
        subprocess.run(['make', '-C', os.path.join(ROOT, 'efficient-nerf_amd', 'csrc'), 'example'], check=True)
    n_block, H, W = 2, 24, 40  # non-square on purpose
    focal = O.focal_from_angle(W)
    sd = O.make_r2l_state(seed=8, netdepth=2 + 2 * n_block)
    names = O.r2l_state_names(n_block)
    np.concatenate([sd[n].float().numpy().ravel() for n in names]).astype(np.float32).tofile(tmp_path / 'w.bin')
    c2w = O.pose_spherical(40., -25., 4.)[:3, :4].contiguous()
    c2w.numpy().astype(np.float32).tofile(tmp_path / 'pose.bin')
    r = subprocess.run([exe, str(tmp_path / 'w.bin'), str(tmp_path / 'pose.bin'), str(tmp_path / 'out.bin'), str(H), str(W),
                        repr(float(focal)), str(n_block), str(precision)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f'rendered {H}x{W}' in r.stdout
    rgb = np.fromfile(tmp_path / 'out.bin', dtype=np.float32).reshape(H * W, 3)
    ref = O.r2l_render(sd, H, W, focal, c2w).numpy()
    err = np.abs(rgb - ref).max()
    print(f'C client precision={precision}: L_inf vs CPU oracle {err:.2e}')
    assert err <= tol


def test_c_client_reports_errors(built_lib, tmp_path):
    exe = os.path.join(ROOT, 'examples', 'r2l_render')
    if not os.path.exists(exe):
        pytest.skip('example not built')
    np.zeros(10, np.float32).tofile(tmp_path / 'short.bin')
    np.zeros(12, np.float32).tofile(tmp_path / 'pose.bin')
    r = subprocess.run([exe, str(tmp_path / 'short.bin'), str(tmp_path / 'pose.bin'), str(tmp_path / 'o.bin'), '8', '8', '10.0', '1'],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and 'expected' in r.stderr
