"""The generated gfx950 head layer of the R2L network (csrc/gen/head_gen.py), checked on the CPU: the generator's
lane-accurate emulator runs the exact instruction stream that is assembled into r2l_head_kernel -- embedding arithmetic
included -- on the bytes the C++ packer (r2l_capi.hip pack_head_v1) produces, and relu(head) of a wave's 32 rays is
compared with a float64 evaluation of model/nerf_raybased.py:94-102 (sample_test), :191-208 (PositionalEmbedder) and
the head Linear(1008, 256) + ReLU (:539-541)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'efficient-nerf_amd', 'csrc', 'gen'))
import head_gen as G  # noqa: E402

import _pkg  # noqa: E402
_pkg.load()
from efficient_nerf_amd import _lib  # noqa: E402
from oracle import r2l_oracle as O  # noqa: E402


def cxx_pack(sd, n_block, fmt='bf6'):
    G.configure(fmt)
    mode = {'bf6': 2, 'f16': 4}[fmt]       # R2L_PREC_FP16_FP8 | R2L_PREC_FP16X3_ASM
    keep, arr = _lib.host_ptrs([sd[n] for n in O.r2l_state_names(n_block)])
    n = _lib.lib().r2l_debug_pack_host(arr, len(keep), n_block, mode, None, 0)
    assert n == G.STREAM_BYTES + G.AUX_BYTES, (n, _lib.lib().r2l_last_error())
    buf = np.zeros(n, dtype=np.uint8)
    assert _lib.lib().r2l_debug_pack_host(arr, len(keep), n_block, mode, C.c_void_p(buf.ctypes.data), n) == n
    return buf


def test_head_columns_are_a_bijection():
    cols = [G.head_col(p, s, h, j) for p in range(16) for s in range(4) for h in range(2) for j in range(8)]
    assert sorted(c for c in cols if c >= 0) == list(range(1008)) and cols.count(-1) == 16


@pytest.mark.parametrize('fmt', ['bf6', 'f16'])
def test_cxx_packer_matches_python_restatement(fmt):
    sd = O.make_r2l_state(seed=5, netdepth=4)
    buf = cxx_pack(sd, 1, fmt)
    img, aux = G.pack_head(sd['head.0.weight'].numpy(), sd['head.0.bias'].numpy(), fmt=fmt)
    assert img.size == 32 * {'bf6': 28672, 'f16': 32768}[fmt]
    assert np.array_equal(buf[:G.STREAM_BYTES], img)
    if fmt == 'f16':      # the scale words of the aux block are not read by the three-pass build
        assert np.array_equal(buf[G.STREAM_BYTES:G.STREAM_BYTES + 1024], aux[:1024])
    else:
        assert np.array_equal(buf[G.STREAM_BYTES:], aux)
    G.configure('bf6')


@pytest.mark.parametrize('fmt', ['bf6', 'f16'])
def test_committed_asm_is_the_generators_output(tmp_path, fmt):
    G.emit(str(tmp_path), G.Opts(fmt=fmt))
    stem = {'bf6': 'r2l_head', 'f16': 'r2l_headx'}[fmt]
    for name in ('_asm.inc', '_pro_asm.inc', '_clobbers.inc', '_pro_clobbers.inc'):
        built = os.path.join(ROOT, 'efficient-nerf_amd', 'csrc', stem + name)
        assert open(os.path.join(str(tmp_path), stem + name)).read() == open(built).read(), stem + name
    G.configure('bf6')


@pytest.mark.parametrize('wave,n_tiles,first,fmt', [(0, 1, 0, 'bf6'), (3, 2, 100, 'bf6'), (1, 1, 40, 'f16'), (2, 2, 200, 'f16')])
def test_emulated_head_vs_float64(wave, n_tiles, first, fmt):
    sd = O.make_r2l_state(seed=wave + 1, netdepth=4)
    W, b = sd['head.0.weight'].numpy(), sd['head.0.bias'].numpy()
    H = 16
    focal = O.focal_from_angle(H)
    c2w = O.pose_spherical(30. + 10 * wave, -30., 4.)
    z = O.sampler_z_vals(16, 2., 6.)
    ro, rd = O.rays_from_dirs(O.camera_dirs(H, H, focal), c2w[:3, :4])
    ro, rd = ro.reshape(-1, 3)[first:first + 32], rd.reshape(-1, 3)[first:first + 32]
    emb = O.positional_embed(O.sample_rays(ro, rd, z), 10).double().numpy()
    ref = np.maximum(emb @ W.astype(np.float64).T + b, 0)
    f16 = lambda a: a.astype(np.float16).astype(np.float64)   # noqa: E731
    f16_err = np.abs(np.maximum(f16(emb) @ f16(W).T + b, 0) - ref).max()
    buf = cxx_pack(sd, 1, fmt)
    lanes = np.arange(64)
    o = [ro[lanes & 31, c].numpy() for c in range(3)]
    d = [rd[lanes & 31, c].numpy() for c in range(3)]
    out, errs = G.emulate_tile(G.Opts(fmt=fmt), buf[:G.STREAM_BYTES], buf[G.STREAM_BYTES:], o, d, z.numpy(), wave=wave, n_tiles=n_tiles)
    G.configure('bf6')
    assert not errs, errs[:10]
    got = np.zeros((32, 256))
    for u in range(8):       # register image: group 4u + g of lane 32h + ray = features 32u + 8g + 4h .. + 3
        for g in range(4):
            for r in range(4):
                got[lanes & 31, 32 * u + 8 * g + 4 * (lanes >> 5) + r] = out[4 * u + g][r]
    err = np.abs(got / 16.0 - ref).max()
    print('wave %d %s: L_inf %.3g (single-pass fp16 operands: %.3g), |h0| max %.3g' % (wave, fmt, err, f16_err, ref.max()))
    # bf6 terms: ~2.5e-5; three fp16 passes: what is left is the embedding's own arithmetic (v_sin_f32, fp32 range reduction)
    assert err < (5e-5 if fmt == 'bf6' else 3e-6) and err < f16_err / 8
