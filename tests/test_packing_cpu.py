"""CPU: host logic of r2l_load_weights — the packed MFMA chunk stream (csrc/r2l_common.h)
decoded in Python with the index maps restated here, against the fp32 weights.  Runs
without a GPU through r2l_debug_pack_host."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import r2l_oracle as O

FRAG, AUXB, FRAGS = 1024, 1024, 16


def kappa(ks, h, j):
    return 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3)


def head_col(ks, h, j):
    if ks < 48:
        return ks * 21 + (10 if h else 0) + j
    if ks < 60:
        return (4 * (ks - 48) + (j >> 1)) * 21 + (10 if h else 0) + 8 + (j & 1)
    if ks < 63:
        return (16 * (ks - 60) + 8 * h + j) * 21 + 20
    return -1


def test_index_maps_are_bijections():
    for ks_pair in range(8):  # every 32-feature group is covered exactly once by two k-steps
        got = sorted(kappa(2 * ks_pair + s, h, j) for s in (0, 1) for h in (0, 1) for j in range(8))
        assert got == list(range(32 * ks_pair, 32 * ks_pair + 32))
    cols = [head_col(ks, h, j) for ks in range(64) for h in (0, 1) for j in range(8)]
    real = sorted(c for c in cols if c >= 0)
    assert real == list(range(1008)) and cols.count(-1) == 16


def pack(pkg, sd, n_block, mode):
    from efficient_nerf_amd import _lib
    names = O.r2l_state_names(n_block)
    keep, arr = _lib.host_ptrs([sd[n] for n in names])
    L = _lib.lib()
    size = L.r2l_debug_pack_host(arr, len(keep), n_block, mode, None, 0)
    assert size > 0
    buf = (C.c_char * size)()
    assert L.r2l_debug_pack_host(arr, len(keep), n_block, mode, buf, size) == size
    return np.frombuffer(buf, dtype=np.uint8).copy()


@pytest.mark.parametrize('mode,np_', [(0, 2), (1, 1)])
def test_packed_stream_decodes_to_weights(pkg, built_lib, mode, np_):
    n_block = 2
    sd = O.make_r2l_state(seed=9, netdepth=2 + 2 * n_block)
    img = pack(pkg, sd, n_block, mode)
    CH = FRAGS * np_ * FRAG + AUXB
    cpt = 32 + 2 * n_block * 8 + 1
    assert img.size == cpt * CH

    def frag(chunk, f, part):
        off = chunk * CH + (f * np_ + part) * FRAG
        return img[off:off + FRAG].view(np.float16).reshape(64, 8).astype(np.float64)

    def aux(chunk):
        off = chunk * CH + FRAGS * np_ * FRAG
        return img[off:off + AUXB].view(np.float32)

    def value(chunk, f):  # hi (+ lo)
        v = frag(chunk, f, 0)
        return v + frag(chunk, f, 1) if np_ == 2 else v

    tol = 2.0 ** -21 if np_ == 2 else 2.0 ** -11  # relative to the scaled max (|w|*S in [2^12, 2^13))

    # body layers: chunk = feature tile t, frag = k-step
    for li in range(2 * n_block):
        Wl = sd[O.r2l_state_names(n_block)[2 + 2 * li]].double().numpy()
        bl = sd[O.r2l_state_names(n_block)[3 + 2 * li]].double().numpy()
        for t in range(8):
            ci = 32 + li * 8 + t
            a = aux(ci)
            inv = float(a[32])
            S = 1.0 / inv
            assert S == 2.0 ** round(np.log2(S))  # power of two
            np.testing.assert_allclose(a[:32] * inv, bl[32 * t:32 * t + 32], rtol=1e-6, atol=1e-9)
            Sw = S / 16.0
            assert 2 ** 12 <= np.abs(Wl).max() * Sw < 2 ** 13
            for ks in (0, 7, 15):
                v = value(ci, ks)
                for lane in (0, 17, 33, 63):
                    want = np.array([Wl[32 * t + (lane & 31), kappa(ks, lane >> 5, j)] for j in range(8)]) * Sw
                    assert np.abs(v[lane] - want).max() <= tol * 2 ** 13
    # head: chunk c = k-steps 2c, 2c+1; frag = ksl*8 + t
    Wh = sd['head.0.weight'].double().numpy()
    inv = float(aux(31)[32])
    Sw = 1.0 / inv / 16.0
    np.testing.assert_allclose(aux(0)[:256] * inv, sd['head.0.bias'].double().numpy(), rtol=1e-6, atol=1e-9)
    for ks in (0, 47, 48, 59, 60, 62, 63):
        for t in (0, 5):
            v = value(ks // 2, (ks & 1) * 8 + t)
            for lane in (3, 40):
                want = np.array([0.0 if head_col(ks, lane >> 5, j) < 0 else
                                 Wh[32 * t + (lane & 31), head_col(ks, lane >> 5, j)] for j in range(8)]) * Sw
                assert np.abs(v[lane] - want).max() <= tol * 2 ** 13
    # tail: rows 0..2 real, the rest zero
    Wt = sd['tail.0.weight'].double().numpy()
    inv = float(aux(cpt - 1)[32])
    Sw = 1.0 / inv / 16.0
    v = value(cpt - 1, 4)
    for lane in (0, 2, 34):
        want = np.array([Wt[lane & 31, kappa(4, lane >> 5, j)] for j in range(8)]) * Sw
        assert np.abs(v[lane] - want).max() <= tol * 2 ** 13
    assert np.abs(v[3:32]).max() == 0 and np.abs(v[35:]).max() == 0


def test_pack_rejects_bad_input(pkg, built_lib):
    from efficient_nerf_amd import _lib
    sd = O.make_r2l_state(seed=1, netdepth=4)
    keep, arr = _lib.host_ptrs([sd[n] for n in O.r2l_state_names(1)])
    L = _lib.lib()
    assert L.r2l_debug_pack_host(arr, len(keep), 2, 0, None, 0) < 0  # tensor count does not match n_block
    assert L.r2l_debug_pack_host(arr, len(keep), 1, 7, None, 0) < 0  # bad precision mode
