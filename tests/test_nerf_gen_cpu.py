"""The generated gfx950 layer chain of the NeRF teacher (csrc/gen/nerf_gen.py), checked on the CPU: the generator's
lane-accurate emulator runs the exact instruction stream that is assembled into nerf_chain_kernel on the bytes the
C++ packer (nerf_capi.hip pack_chain) produces, and raw = (rgb, sigma) of 32 points is compared with a float64
evaluation of NeRF.forward (model/nerf_raybased.py:377-401) with both embedders
(utils/run_nerf_raybased_helpers.py:24-56).  The emulator enforces the stream's contracts (counted lgkmcnt before a
ds_read's data is used, vmcnt + barrier before LDS-DMA bytes are read) and body_gen's static hazard rules."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'efficient-nerf_amd', 'csrc', 'gen'))
import nerf_gen as G  # noqa: E402

import _pkg  # noqa: E402
_pkg.load()
from efficient_nerf_amd import _lib  # noqa: E402

SHAPES = [(256, 63), (256,), (256, 256), (256,), (256, 256), (256,), (256, 256), (256,), (256, 256), (256,), (256, 319), (256,),
          (256, 256), (256,), (256, 256), (256,), (128, 283), (128,), (256, 256), (256,), (1, 256), (1,), (3, 128), (3,)]


def make_tensors(seed=0, gain=1.0):
    """the 24 state_dict tensors with nn.Linear's default init (uniform +- 1/sqrt(fan_in)), times gain"""
    rng = np.random.default_rng(seed)
    out = []
    for i, s in enumerate(SHAPES):
        b = gain / np.sqrt(SHAPES[i - (i & 1)][1])
        out.append(rng.uniform(-b, b, size=s).astype(np.float32))
    return out


def embed(x, L):
    out = [x]
    for l in range(L):
        out += [np.sin(x * 2.0 ** l), np.cos(x * 2.0 ** l)]
    return np.concatenate(out, -1)


def ref_mlp(t, pts, vd, f16_ops=False):
    """raw [n, 4] in float64; f16_ops: the operands of every product rounded to fp16 (a single-pass fp16 kernel)"""
    r = (lambda a: a.astype(np.float16).astype(np.float64)) if f16_ops else (lambda a: a)
    t = [x.astype(np.float64) for x in t]
    e, v = embed(pts.astype(np.float64), 10), embed(vd.astype(np.float64), 4)
    h = e
    for i in range(8):
        h = np.maximum(r(h) @ r(t[2 * i]).T + t[2 * i + 1], 0)
        if i == 4:
            h = np.concatenate([e, h], -1)
    alpha = r(h) @ r(t[20]).T + t[21]
    feat = r(h) @ r(t[18]).T + t[19]
    hv = np.maximum(r(np.concatenate([feat, v], -1)) @ r(t[16]).T + t[17], 0)
    return np.concatenate([r(hv) @ r(t[22]).T + t[23], alpha], -1), e, v


def make_frags(e, v, act):
    """input operands of one wave: e [32, 63], v [32, 27] -> {name: uint32 [4, 64]} (hi | lo fp16 fragments)"""
    lanes = np.arange(64)
    q, n = lanes >> 4, lanes & 15
    fr = {}
    for kind, ne, src in (('E', 2, e), ('V', 1, v)):
        for ee in range(ne):
            for c in range(2):
                H = np.zeros((64, 8), np.float16)
                Lo = np.zeros((64, 8), np.float16)
                for j in range(8):
                    for l in range(64):
                        col = G.pts_col(ee, q[l], j) if kind == 'E' else G.view_col(q[l], j)
                        if col < 0:
                            continue
                        val = np.float32(src[c * 16 + n[l], col]) * np.float32(act)
                        H[l, j] = np.float16(val)
                        Lo[l, j] = np.float16(val - np.float32(H[l, j]))
                hn, ln = (('eh%d%d' % (ee, c), 'el%d%d' % (ee, c)) if kind == 'E' else ('vh%d' % c, 'vl%d' % c))
                fr[hn] = H.view(np.uint32).T.copy()
                fr[ln] = Lo.view(np.uint32).T.copy()
    return fr


def cxx_pack(tensors):
    keep, arr = _lib.host_ptrs([torch.from_numpy(np.ascontiguousarray(t)) for t in tensors])
    offs = (C.c_longlong * 1)()
    n = _lib.lib().nerf_debug_pack_chain_host(arr, len(keep), 0, None, 0, offs)
    assert n > 0, _lib.lib().r2l_last_error()
    buf = np.zeros(n, dtype=np.uint8)
    assert _lib.lib().nerf_debug_pack_chain_host(arr, len(keep), 0, C.c_void_p(buf.ctypes.data), n, offs) == n
    return buf, int(offs[0])


def test_layout_constants():
    assert G.NCH % G.NSLOT == 0 and G.N_ANCH == 3732 and G.NT == 154
    assert G.STREAM_BYTES == 2166784          # NERF_CHAIN_STREAM_BYTES (csrc/nerf_common.h)
    assert (G.AUX_BYTES, G.AUX_LAYER, G.AUX_SCALES, G.SLOT * G.NSLOT) == (16384, 1280, 1152, 131072)
    assert 11 * G.AUX_LAYER <= G.AUX_BYTES


def test_cxx_packer_matches_python_restatement():
    t = make_tensors(seed=3)
    buf, aux_off = cxx_pack(t)
    img, aux = G.pack_teacher(t)
    assert aux_off == img.size and buf.size == img.size + aux.size
    assert np.array_equal(buf[:aux_off], img)
    assert np.array_equal(buf[aux_off:], aux)


def test_committed_asm_is_the_generators_output(tmp_path):
    G.emit(str(tmp_path), G.Opts())
    for name in ('nerf_mlp_asm.inc', 'nerf_mlp_pro_asm.inc', 'nerf_mlp_clobbers.inc', 'nerf_mlp_pro_clobbers.inc'):
        built = os.path.join(ROOT, 'efficient-nerf_amd', 'csrc', name)
        assert open(os.path.join(str(tmp_path), name)).read() == open(built).read(), name


@pytest.mark.parametrize('wave,n_tiles,gain', [(0, 1, 1.0), (3, 2, 1.0), (1, 1, 1.5)])
def test_emulated_chain_vs_float64(wave, n_tiles, gain):
    t = make_tensors(seed=wave, gain=gain)
    rng = np.random.default_rng(10 + wave)
    pts = rng.uniform(-2.5, 2.5, size=(32, 3)).astype(np.float32)
    vd = rng.normal(size=(32, 3))
    vd = (vd / np.linalg.norm(vd, axis=1, keepdims=True)).astype(np.float32)
    ref, e, v = ref_mlp(t, pts, vd)
    f16_err = np.abs(ref_mlp(t, pts, vd, f16_ops=True)[0] - ref).max()
    buf, aux_off = cxx_pack(t)
    out, errs = G.emulate_tile(G.Opts(), buf[:aux_off], buf[aux_off:], make_frags(e, v, 16.0), wave=wave, n_tiles=n_tiles)
    assert not errs, errs[:10]
    got = np.zeros((32, 4))
    for c in range(2):
        for k in range(4):
            got[c * 16:(c + 1) * 16, k] = out[c * 4 + k][:16] / 16.0
    err = np.abs(got - ref).max()
    scale = np.abs(ref).max()
    print('wave %d: L_inf %.3g (single-pass fp16 operands: %.3g), |raw| max %.3g' % (wave, err, f16_err, scale))
    assert err < 2e-5 * max(1.0, scale)
    assert err < f16_err / 8


def test_reads_respect_certification_and_buffers():
    """structure of the stream: every chunk's rendezvous precedes the first read of the next chunk, and the counts"""
    body = G.block_stream(G.Opts())
    kinds = {}
    for ins in body:
        kinds[ins.kind] = kinds.get(ins.kind, 0) + 1
    assert kinds['mfma16'] == 2632 and kinds['mfma6'] == 1100 and kinds['barrier'] == G.NCH + 1
    assert kinds['dma'] == sum(G.CHUNKS[(c + 3) % G.NCH]['pw'] for c in range(G.NCH))
