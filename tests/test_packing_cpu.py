"""CPU: host logic of r2l_load_weights — the packed MFMA chunk stream (csrc/r2l_common.h)
decoded in Python with the index maps restated here, against the fp32 weights.  Runs
without a GPU through r2l_debug_pack_host."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import r2l_oracle as O

FRAG, AUXB, FRAGS = 1024, 1024, 16


def kappa(s, q, j):
    return 32 * s + 16 * (j >> 2) + 4 * q + (j & 3)


def head_col(s, q, j):
    if s < 24:
        return (2 * s + (q >> 1)) * 21 + (10 if q & 1 else 0) + j
    if s < 30:
        return (8 * (s - 24) + 2 * q + (j >> 2)) * 21 + (10 if (j >> 1) & 1 else 0) + 8 + (j & 1)
    if s == 30:
        return (8 * q + j) * 21 + 20
    return (32 + 8 * q + j) * 21 + 20 if q < 2 else -1


def test_index_maps_are_bijections():
    for s in range(8):  # every 32-feature group is covered exactly once by one k-step
        got = sorted(kappa(s, q, j) for q in range(4) for j in range(8))
        assert got == list(range(32 * s, 32 * s + 32))
    cols = [head_col(s, q, j) for s in range(32) for q in range(4) for j in range(8)]
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

    # body layers: chunk m = row tiles 2m, 2m+1; frag = (u&1)*8 + k-step
    for li in range(2 * n_block):
        Wl = sd[O.r2l_state_names(n_block)[2 + 2 * li]].double().numpy()
        bl = sd[O.r2l_state_names(n_block)[3 + 2 * li]].double().numpy()
        for m in range(8):
            ci = 32 + li * 8 + m
            a = aux(ci)
            inv = float(a[32])
            S = 1.0 / inv
            assert S == 2.0 ** round(np.log2(S))  # power of two
            np.testing.assert_allclose(a[:32] * inv, bl[32 * m:32 * m + 32], rtol=1e-6, atol=1e-9)
            Sw = S / 16.0
            assert 2 ** 12 <= np.abs(Wl).max() * Sw < 2 ** 13
            for f in (0, 7, 9, 15):
                u, ks = 2 * m + (f >> 3), f & 7
                v = value(ci, f)
                for lane in (0, 17, 33, 63):
                    want = np.array([Wl[16 * u + (lane & 15), kappa(ks, lane >> 4, j)] for j in range(8)]) * Sw
                    assert np.abs(v[lane] - want).max() <= tol * 2 ** 13
    # head: chunk = k-step; frag = row tile u
    Wh = sd['head.0.weight'].double().numpy()
    inv = float(aux(31)[32])
    Sw = 1.0 / inv / 16.0
    np.testing.assert_allclose(aux(0)[:256] * inv, sd['head.0.bias'].double().numpy(), rtol=1e-6, atol=1e-9)
    for ks in (0, 23, 24, 29, 30, 31):
        for u in (0, 5, 15):
            v = value(ks, u)
            for lane in (3, 20, 40, 57):
                want = np.array([0.0 if head_col(ks, lane >> 4, j) < 0 else
                                 Wh[16 * u + (lane & 15), head_col(ks, lane >> 4, j)] for j in range(8)]) * Sw
                assert np.abs(v[lane] - want).max() <= tol * 2 ** 13
    # tail: rows 0..2 of row tile 0 real (frags 0..7), the rest zero
    Wt = sd['tail.0.weight'].double().numpy()
    inv = float(aux(cpt - 1)[32])
    Sw = 1.0 / inv / 16.0
    v = value(cpt - 1, 4)
    for lane in (0, 2, 18, 50):
        want = np.array([Wt[lane & 15, kappa(4, lane >> 4, j)] for j in range(8)]) * Sw
        assert np.abs(v[lane] - want).max() <= tol * 2 ** 13
    assert np.abs(v[3:16]).max() == 0 and np.abs(v[19:32]).max() == 0
    assert np.abs(value(cpt - 1, 9)).max() == 0


def test_fp16_fp8_head_image_and_body_stream(pkg, built_lib):
    """R2L_PREC_FP16_FP8 streams two images: the head launch reads the stream of r2l_debug_pack_host(mode 2) (pinned
    against the generator's Python restatement in tests/test_head_gen_cpu.py), the body kernel the 28 KiB-chunk stream of
    r2l_debug_pack_body_host (16 fp16 fragments + 8 bf6 operands; pinned in tests/test_body_gen_cpu.py).  Here: sizes,
    scale bytes, and the bf6 codes decode to the weights they stand for."""
    import ctypes as C
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'efficient-nerf_amd', 'csrc', 'gen'))
    import body_gen as G
    from efficient_nerf_amd import _lib
    n_block = 1
    sd = O.make_r2l_state(seed=4, netdepth=2 + 2 * n_block)
    assert pack(pkg, sd, n_block, 2).size == 32 * 28672 + 2048     # mode 2 of r2l_debug_pack_host = the head launch's image
    keep, arr = _lib.host_ptrs([sd[n] for n in O.r2l_state_names(n_block)])
    offs = (C.c_longlong * 2)()
    n = _lib.lib().r2l_debug_pack_body_host(arr, len(keep), n_block, None, 0, offs)
    buf = np.zeros(n, dtype=np.uint8)
    assert _lib.lib().r2l_debug_pack_body_host(arr, len(keep), n_block, C.c_void_p(buf.ctypes.data), n, offs) == n
    G.configure('bf6')
    assert G.CHUNK == 28672 and offs[0] == n_block * 16 * G.CHUNK and offs[1] == offs[0] + n_block * G.AUX_BYTES
    W1 = sd['body.0.body.0.weight'].float().numpy()
    ex = G.layer_exponent(W1)
    el, ew = G.weight_exps(ex)
    aux = buf[offs[0]:offs[0] + G.AUX_BYTES].view(np.uint32)
    assert (aux[256] & 0xff) == 127 + el and (aux[257] & 0xff) == 127 + ew
    # chunk 3 (row tile 3: features 96..127) of layer 1, operand j = 2 (w - hi(w), K=64 step 1): decode lane 37
    import isa
    u, j, lane = 3, 2, 37
    base = u * G.CHUNK
    o1, o2 = G.off_a6(j)
    lo = buf[base + o1 + lane * 16:][:16]
    hi = buf[base + o2 + lane * 8:][:8]
    G.configure('bf6')
    codes = isa.unpack6(np.concatenate([lo, hi]).view(np.uint32)[None])[0]
    vals = isa.BF6[codes] * 2.0 ** el
    w = np.array([W1[32 * u + (lane & 31), G.mix_feat(1, lane >> 5, e)] for e in range(32)])
    want = w.astype(np.float64) - w.astype(np.float16).astype(np.float64)
    assert np.abs(vals - want).max() <= 0.13 * np.abs(want).max()  # e3m2: 2 mantissa bits
    assert np.abs(vals - want).max() > 0


def test_pack_rejects_bad_input(pkg, built_lib):
    from efficient_nerf_amd import _lib
    sd = O.make_r2l_state(seed=1, netdepth=4)
    keep, arr = _lib.host_ptrs([sd[n] for n in O.r2l_state_names(1)])
    L = _lib.lib()
    assert L.r2l_debug_pack_host(arr, len(keep), 2, 0, None, 0) < 0  # tensor count does not match n_block
    assert L.r2l_debug_pack_host(arr, len(keep), 1, 7, None, 0) < 0  # bad precision mode


def test_fp16_fp8_packers_reject_weights_outside_the_split_range(pkg, built_lib):
    """The fp16 + bf6 split covers layers with max|w| in [2^-13, 2^6): beyond it the fp16 residual or the bf6 shift would
    leave their formats, and both packers must say so instead of packing garbage (the caller falls back to fp16x3)."""
    import ctypes as C
    from efficient_nerf_amd import _lib
    L = _lib.lib()
    sd = O.make_r2l_state(seed=2, netdepth=4)
    sd['body.0.body.2.weight'] = sd['body.0.body.2.weight'] * 2.0 ** 12
    keep, arr = _lib.host_ptrs([sd[n] for n in O.r2l_state_names(1)])
    offs = (C.c_longlong * 2)()
    assert L.r2l_debug_pack_body_host(arr, len(keep), 1, None, 0, offs) < 0
    assert b'outside the range' in L.r2l_last_error()
    from efficient_nerf_amd.teacher import NeRFEngine
    t = O.make_teacher_state(1)
    t['pts_linears.3.weight'] = t['pts_linears.3.weight'] * 2.0 ** -12
    keep, arr = _lib.host_ptrs([t[n] for n in NeRFEngine.STATE_NAMES])
    assert L.nerf_debug_pack_chain_host(arr, len(keep), 0, None, 0, offs) < 0
    assert b'teacher layer 3' in L.r2l_last_error() and b'outside the range' in L.r2l_last_error()
    keep, arr = _lib.host_ptrs([O.make_teacher_state(1)[n] for n in NeRFEngine.STATE_NAMES])
    assert L.nerf_debug_pack_chain_host(arr, len(keep), 0, None, 0, offs) > 0


def _fnv(b):
    h = 1469598103934665603
    for c in np.frombuffer(b, dtype=np.uint8).tolist():
        h = ((h ^ c) * 1099511628211) & 0xffffffffffffffff
    return '%016x' % h


def test_host_packers_under_asan_ubsan(pkg, built_lib, tmp_path):
    """the host-only packers (pack_image_host / pack_head_v1 / pack_body_v3 / pack_chain) and the numpy
    shuffle, compiled from the product sources with -fsanitize=address,undefined (`make asan`: host side only, no device
    code) and run on seeded weights: a clean sanitizer log, and byte-for-byte the streams the product library packs."""
    import os
    import subprocess
    from efficient_nerf_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(['make', '-C', os.path.join(root, 'efficient-nerf_amd', 'csrc'), 'asan'], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    n_block = 2
    sd = O.make_r2l_state(seed=21, netdepth=2 + 2 * n_block)
    sd['body.1.body.0.weight'] = sd['body.1.body.0.weight'] * 3.7         # layers with different exponents
    sd['body.0.body.2.weight'] = sd['body.0.body.2.weight'] * 0.11
    tsd = O.make_teacher_state(5)
    names = O.r2l_state_names(n_block)
    tnames = [f'pts_linears.{i}.{k}' for i in range(8) for k in ('weight', 'bias')] + [
        'views_linears.0.weight', 'views_linears.0.bias', 'feature_linear.weight', 'feature_linear.bias',
        'alpha_linear.weight', 'alpha_linear.bias', 'rgb_linear.weight', 'rgb_linear.bias']
    path = str(tmp_path / 'weights.bin')
    with open(path, 'wb') as f:
        f.write(np.int32(n_block).tobytes())
        for n in names:
            f.write(sd[n].contiguous().numpy().astype(np.float32).tobytes())
        for n in tnames:
            f.write(tsd[n].contiguous().numpy().astype(np.float32).tobytes())
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1')
    r = subprocess.run([os.path.join(root, 'tests', '_build', 'pack_asan'), path], capture_output=True, text=True, env=env, timeout=600)
    log = r.stdout + r.stderr
    assert r.returncode == 0 and 'AddressSanitizer' not in log and 'runtime error' not in log and 'LeakSanitizer' not in log, log[-3000:]
    got = {}
    for ln in r.stdout.splitlines():
        t = ln.split()
        got[(t[0], t[1])] = t[2:]
    # the product library on the same tensors
    L = _lib.lib()
    keep, arr = _lib.host_ptrs([sd[n] for n in names])
    for mode in range(7):         # ... incl. the two-part modes fp16_split / fp16_split8
        size = L.r2l_debug_pack_host(arr, len(keep), n_block, mode, None, 0)
        buf = (C.c_char * size)()
        assert L.r2l_debug_pack_host(arr, len(keep), n_block, mode, buf, size) == size
        assert got[('pack_host', str(mode))] == [str(size), _fnv(bytes(buf))], mode
    for fmt in (0, 1, 3):
        assert L.r2l_debug_pack_body_format(fmt) == 0
        offs = (C.c_longlong * 2)()
        size = L.r2l_debug_pack_body_host(arr, len(keep), n_block, None, 0, offs)
        buf = (C.c_char * size)()
        assert L.r2l_debug_pack_body_host(arr, len(keep), n_block, buf, size, offs) == size
        assert got[('pack_body', str(fmt))] == [str(size), _fnv(bytes(buf)), str(offs[0]), str(offs[1])], fmt
    L.r2l_debug_pack_body_format(0)
    tkeep, tarr = _lib.host_ptrs([tsd[n] for n in tnames])
    for fmt in (0, 1, 2, 3, 4, 5, 6):   # (5, 6: with the second exit) the chain streams: bf6 terms, one fp16 pass, three passes (the p3 hi | lo layout: ADVICE r5), mix (round 6)
        off = (C.c_longlong * 1)()
        size = L.nerf_debug_pack_chain_host(tarr, 24, fmt, None, 0, off)
        buf = (C.c_char * size)()
        assert L.nerf_debug_pack_chain_host(tarr, 24, fmt, buf, size, off) == size
        assert got[('pack_chain', str(fmt))] == [str(size), _fnv(bytes(buf)), str(off[0])], fmt
    key = (np.uint32(2654435761) * np.arange(1, 625, dtype=np.uint32)).astype(np.uint32)
    pos = C.c_int(300)
    for n in (0, 1, 2, 63, 64, 65, 100000):
        out = np.empty(n, dtype=np.int32)
        assert L.r2l_np_legacy_permutation(key.ctypes.data_as(C.c_void_p), C.byref(pos), n, out.ctypes.data_as(C.c_void_p)) == 0
        assert got[('perm', str(n))] == [_fnv(out.tobytes()), str(pos.value)], n
