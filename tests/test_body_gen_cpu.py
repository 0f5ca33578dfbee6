"""The generated gfx950 body loop (csrc/gen/body_gen.py), checked on the CPU: the generator's lane-accurate
emulator runs the exact instruction stream that is assembled into r2l_body_kernel on the bytes the C++
packer (pack_body_v3) produces, and the result is compared with a float64 evaluation of the ResMLP blocks
(model/nerf_raybased.py:443-465).  The emulator also enforces the stream's own contracts: no register is
read before the counted lgkmcnt covers its ds_read, no LDS byte is read before its LDS-DMA was certified
by vmcnt + barrier, MFMA results are not touched too early."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'efficient-nerf_amd', 'csrc', 'gen'))
import body_gen as G  # noqa: E402

import _pkg  # noqa: E402
_pkg.load()
from efficient_nerf_amd import _lib  # noqa: E402


def make_weights(nb, seed=0, gain=1.0):
    rng = np.random.default_rng(seed)
    k = 1 / 16
    W1s = [(gain * rng.uniform(-k, k, (256, 256))).astype(np.float32) for _ in range(nb)]
    W2s = [rng.uniform(-k, k, (256, 256)).astype(np.float32) for _ in range(nb)]
    b1s = [rng.uniform(-k, k, 256).astype(np.float32) for _ in range(nb)]
    b2s = [rng.uniform(-k, k, 256).astype(np.float32) for _ in range(nb)]
    return W1s, b1s, W2s, b2s


def ref_blocks(x, W1s, b1s, W2s, b2s):
    x = x.astype(np.float64)
    for W1, b1, W2, b2 in zip(W1s, b1s, W2s, b2s):
        h = np.maximum(x @ W1.astype(np.float64).T + b1, 0)
        x = x + h @ W2.astype(np.float64).T + b2
    return x


def fp16x1_error(x, W1s, b1s, W2s, b2s):
    """L_inf of the same blocks with fp16-rounded operands and exact accumulation (what the correction terms
    are there to remove)"""
    def f16(a):
        return a.astype(np.float16).astype(np.float64)
    xx = x.astype(np.float64)
    for W1, b1, W2, b2 in zip(W1s, b1s, W2s, b2s):
        h = np.maximum(f16(xx) @ f16(W1).T + b1, 0)
        xx = xx + f16(h) @ f16(W2).T + b2
    return np.abs(xx - ref_blocks(x, W1s, b1s, W2s, b2s)).max()


def to_regs(xs):
    """[32 rays, 256] -> the wave's register image [128, 64]: register 16u + r of lane 32h + ray = feature
    32u + 8(r/4) + 4h + r%4"""
    regs = np.zeros((128, 64), dtype=np.float32)
    lanes = np.arange(64)
    for u in range(8):
        for r in range(16):
            regs[u * 16 + r] = xs[lanes & 31, 32 * u + 8 * (r // 4) + 4 * (lanes >> 5) + r % 4]
    return regs


def from_regs(regs):
    xs = np.zeros((32, 256))
    lanes = np.arange(64)
    for u in range(8):
        for r in range(16):
            xs[lanes & 31, 32 * u + 8 * (r // 4) + 4 * (lanes >> 5) + r % 4] = regs[u * 16 + r]
    return xs


def cxx_pack(W1s, b1s, W2s, b2s, e4m3=0):
    """the C++ packer's stream: 0 = bf6 terms, all operands streamed (the shipping form), 1 = e4m3 terms, 2 = the bf6r
    experiment (22 KiB chunks)"""
    nb = len(W1s)
    _lib.lib().r2l_debug_pack_body_format(e4m3)
    tensors = [np.zeros((256, 1008), np.float32), np.zeros(256, np.float32)]
    for b in range(nb):
        tensors += [W1s[b], b1s[b], W2s[b], b2s[b]]
    rng = np.random.default_rng(5)
    tensors += [rng.uniform(-1 / 16, 1 / 16, (3, 256)).astype(np.float32), rng.uniform(-1 / 16, 1 / 16, 3).astype(np.float32)]
    keep, arr = _lib.host_ptrs([torch.from_numpy(np.ascontiguousarray(t)) for t in tensors])
    offs = (C.c_longlong * 2)()
    n = _lib.lib().r2l_debug_pack_body_host(arr, len(keep), nb, None, 0, offs)
    assert n > 0, _lib.lib().r2l_last_error()
    buf = np.zeros(n, dtype=np.uint8)
    assert _lib.lib().r2l_debug_pack_body_host(arr, len(keep), nb, C.c_void_p(buf.ctypes.data), n, offs) == n
    _lib.lib().r2l_debug_pack_body_format(0)
    return buf, int(offs[0]), int(offs[1]), tensors


def test_cxx_packer_matches_python_restatement():
    W = make_weights(2, seed=3)
    buf, aux_off, tail_off, tensors = cxx_pack(*W)
    img, aux, Bsum = G.pack_body_image(*W)
    assert aux_off == img.size and tail_off == aux_off + 2 * G.AUX_BYTES
    assert np.array_equal(buf[:aux_off], img)
    assert np.array_equal(buf[aux_off:tail_off].view(np.uint32).reshape(2, -1), aux)
    tw = buf[tail_off:].view(np.float32)
    Wt, bt = tensors[-2], tensors[-1]
    assert np.allclose(tw[:768].reshape(3, 256) * 16.0, Wt, rtol=0, atol=0)
    assert np.allclose(tw[768:771], bt + Wt.astype(np.float64) @ Bsum, rtol=1e-6, atol=1e-7)


def test_cxx_packer_matches_python_restatement_e4m3():
    W = make_weights(2, seed=5, gain=3.0)
    buf, aux_off, tail_off, _ = cxx_pack(*W, e4m3=1)
    img, aux, _ = G.pack_body_image(*W, fmt='fp8')
    G.configure('bf6')
    assert aux_off == img.size == 2 * 16 * 32768 and tail_off == aux_off + 2 * G.AUX_BYTES
    bw, aw, tw, _ = cxx_pack(*W, e4m3=2)            # the bf6r experiment's stream
    imgw, auxw, _ = G.pack_body_image(*W, fmt='bf6r')
    G.configure('bf6')
    assert aw == imgw.size == 2 * 16 * 22528 and np.array_equal(bw[:aw], imgw)
    assert np.array_equal(bw[aw:tw].view(np.uint32).reshape(2, -1), auxw)
    assert np.array_equal(buf[:aux_off], img)
    assert np.array_equal(buf[aux_off:tail_off].view(np.uint32).reshape(2, -1), aux)


@pytest.mark.parametrize('nb,wave,burst', [(1, 0, False), (3, 2, False), (2, 3, True)])
def test_emulated_stream_matches_float64(nb, wave, burst):
    W = make_weights(nb, seed=nb)
    buf, aux_off, tail_off, _ = cxx_pack(*W)
    img = buf[:aux_off]
    aux = buf[aux_off:tail_off].view(np.uint32).reshape(nb, -1)
    _, _, Bsum = G.pack_body_image(*W) if nb == 1 else (None, None, np.sum([b.astype(np.float64) for b in W[3][:nb]], axis=0))
    rng = np.random.default_rng(7)
    x = np.maximum(rng.normal(0, 1, (32, 256)), 0).astype(np.float32)
    S = 16.0
    out, errs = G.emulate_tile(G.Opts(dma_burst=burst), img, aux, to_regs(x * S), nb, wave=wave)
    assert not errs, errs[:10]
    got = from_regs(out) / S + Bsum
    ref = ref_blocks(x, *W)
    err = np.abs(got - ref).max()
    # fp16 main pass + bf6 x bf6 correction terms: ~1e-5 per block at |x| ~ 4 (plain fp16: 3e-4)
    assert err < 2.5e-5 * nb and err < fp16x1_error(x, *W) / 8, err


def test_layer_scales_follow_the_weight_exponent():
    # a layer 8x larger needs other E8M0 scales; the stream must stay as accurate
    W = make_weights(1, seed=11, gain=8.0)
    buf, aux_off, tail_off, _ = cxx_pack(*W)
    aux = buf[aux_off:tail_off].view(np.uint32).reshape(1, -1)
    assert (aux[0, 256] & 0xff) == 127 + G.weight_exps(G.layer_exponent(W[0][0]))[0]
    assert (aux[0, 258] & 0xff) == 127 + G.weight_exps(G.layer_exponent(W[2][0]))[0]
    x = np.maximum(np.random.default_rng(1).normal(0, 1, (32, 256)), 0).astype(np.float32)
    out, errs = G.emulate_tile(G.Opts(), buf[:aux_off], aux, to_regs(x * 16.0), 1)
    assert not errs, errs[:10]
    ref = ref_blocks(x, *W)
    got = from_regs(out) / 16.0 + W[3][0].astype(np.float64)
    assert np.abs(got - ref).max() < fp16x1_error(x, *W) / 8


def test_emulated_stream_with_calibrated_exponents():
    """Per-set activation exponents (aux block, AUX_ACT): hidden activations 2^-5 smaller and a residual stream 2^3
    larger than the defaults assume.  With the default exponent 3 the bf6 terms under- / overflow; with exponents
    matched to the ranges the stream is as accurate as on O(1) data."""
    nb, wave = 2, 1
    W1s, b1s, W2s, b2s = make_weights(nb, seed=8)
    k = -5
    W1s = [w * np.float32(2.0 ** k) for w in W1s]
    b1s = [b * np.float32(2.0 ** k) for b in b1s]
    W2s = [w * np.float32(2.0 ** -k) for w in W2s]
    W = (W1s, b1s, W2s, b2s)
    rng = np.random.default_rng(9)
    x = (8.0 * np.maximum(rng.normal(0, 1, (32, 256)), 0)).astype(np.float32)
    S = 16.0
    ref = ref_blocks(x, *W)
    Bsum = np.sum([b.astype(np.float64) for b in b2s], axis=0)
    errs = {}
    for name, act in (('default', None), ('matched', [6, -1, 6, -1, 6])):
        img, aux, _ = G.pack_body_image(*W, act=act)
        out, e = G.emulate_tile(G.Opts(), img, aux, to_regs(x * S), nb, wave=wave)
        assert not e, e[:10]
        errs[name] = np.abs(from_regs(out) / S + Bsum - ref).max()
    print('L_inf default exponents %.3g, matched %.3g (|x| up to %.1f)' % (errs['default'], errs['matched'], np.abs(ref).max()))
    assert errs['matched'] < 2e-5 * np.abs(ref).max()
    assert errs['matched'] < errs['default'] / 2


def test_range_guard_stream_collects_the_maxima_of_every_operand_set():
    """The guard build of the stream (r2l_body_guard_kernel): same results bit for bit, plus one row of per-lane maxima
    of |a| per operand set (IN_b = the folded residual stream at block b, H_b = its hidden layer) in LDS -- what the
    library compares with the calibrated bf6 exponents (r2l_get_range_status)."""
    nb, wave = 3, 1
    W = make_weights(nb, seed=4)
    buf, aux_off, tail_off, _ = cxx_pack(*W)
    img = buf[:aux_off]
    aux = buf[aux_off:tail_off].view(np.uint32).reshape(nb, -1)
    rng = np.random.default_rng(12)
    x = np.maximum(rng.normal(0, 1.5, (32, 256)), 0).astype(np.float32)
    S = 16.0
    plain, e0 = G.emulate_tile(G.Opts(), img, aux, to_regs(x * S), nb, wave=wave)
    out, e1, rows = G.emulate_tile(G.Opts(guard=True), img, aux, to_regs(x * S), nb, wave=wave)
    assert not e0 and not e1, (e0 + e1)[:10]
    assert np.array_equal(plain.view(np.uint32), out.view(np.uint32))
    # float64 evaluation of the folded blocks: x~_b = x_b - sum_{j<b} b2_j
    xs = x.astype(np.float64)
    want = []
    for W1, b1, W2, b2 in zip(*W):
        want.append(np.abs(xs).max())
        h = np.maximum((xs @ W1.astype(np.float64).T) + b1, 0)      # b1' applied to x~ equals b1 applied to x
        want.append(h.max())
        xs = xs + h @ W2.astype(np.float64).T + b2
    b2sum = np.zeros(256)
    got = rows.max(axis=1) / S
    for b in range(nb):      # the stream carries x~: compare in x~ units
        xt = np.abs(ref_blocks(x, *[w[:b] for w in W]) - b2sum).max() if b else np.abs(x).max()
        assert abs(got[2 * b] - xt) <= 2e-3 * xt, (b, got[2 * b], xt)
        assert abs(got[2 * b + 1] - want[2 * b + 1]) <= 2e-3 * want[2 * b + 1], (b, got[2 * b + 1], want[2 * b + 1])
        b2sum = b2sum + W[3][b].astype(np.float64)


def test_e4m3_stream_matches_float64_at_half_the_bf6_error():
    """The middle precision mode (R2L_PREC_FP16_E4M3): the same machine with both correction terms in OCP e4m3 (8-register
    operands, 32 KiB chunks, conversions by v_cvt_scalef32_pk_fp8_f16): one more mantissa bit in all four factors."""
    nb, wave = 2, 2
    W = make_weights(nb, seed=6)
    rng = np.random.default_rng(3)
    x = np.maximum(rng.normal(0, 1, (32, 256)), 0).astype(np.float32)
    S = 16.0
    ref = ref_blocks(x, *W)
    Bsum = np.sum([b.astype(np.float64) for b in W[3]], axis=0)
    errs = {}
    for fmt in ('bf6', 'fp8'):
        img, aux, _ = G.pack_body_image(*W, fmt=fmt)
        out, e = G.emulate_tile(G.Opts(fmt=fmt), img, aux, to_regs(x * S), nb, wave=wave)
        assert not e, (fmt, e[:10])
        errs[fmt] = np.abs(from_regs(out) / S + Bsum - ref).max()
    G.configure('bf6')
    print('L_inf after %d blocks: bf6 terms %.3g, e4m3 terms %.3g' % (nb, errs['bf6'], errs['fp8']))
    assert errs['fp8'] < 0.75 * errs['bf6'] and errs['fp8'] < 1.5e-5 * nb


def test_e4m3_guard_stream_is_bit_identical():
    nb = 2
    W = make_weights(nb, seed=9)
    img, aux, _ = G.pack_body_image(*W, fmt='fp8')
    x = np.maximum(np.random.default_rng(2).normal(0, 1, (32, 256)), 0).astype(np.float32)
    plain, e0 = G.emulate_tile(G.Opts(fmt='fp8'), img, aux, to_regs(x * 16.0), nb, wave=3)
    out, e1, rows = G.emulate_tile(G.Opts(fmt='fp8', guard=True), img, aux, to_regs(x * 16.0), nb, wave=3)
    G.configure('bf6')
    assert not e0 and not e1, (e0 + e1)[:10]
    assert np.array_equal(plain.view(np.uint32), out.view(np.uint32))
    assert abs(rows[0].max() / 16.0 - np.abs(x).max()) < 1e-3 and rows.shape == (2 * nb, 32)


@pytest.mark.parametrize('nb,wave,guard', [(2, 0, False), (3, 3, False), (2, 1, True)])
def test_bf6r_stream_converts_the_weight_operands_from_registers(nb, wave, guard):
    """R2L_PREC_FP16_FP8's stream since round 3 ('bf6r'): the four bf6(W) operands of a chunk are not streamed (22 KiB chunks
    instead of 28); each wave converts one of them from the chunk's fp16 fragments and stores it to LDS behind the chunk's
    existing barrier.  Same arithmetic up to the double rounding bf6(fp16(w)) vs bf6(w)."""
    W = make_weights(nb, seed=20 + nb)
    rng = np.random.default_rng(4)
    x = np.maximum(rng.normal(0, 1, (32, 256)), 0).astype(np.float32)
    S = 16.0
    ref = ref_blocks(x, *W)
    Bsum = np.sum([b.astype(np.float64) for b in W[3]], axis=0)
    img0, aux0, _ = G.pack_body_image(*W, fmt='bf6')
    out0, e0 = G.emulate_tile(G.Opts(), img0, aux0, to_regs(x * S), nb, wave=wave)
    img, aux, _ = G.pack_body_image(*W, fmt='bf6r')
    assert img.size == nb * 16 * 22 * 1024 and np.array_equal(aux, aux0)
    res = G.emulate_tile(G.Opts(fmt='bf6r', guard=guard), img, aux, to_regs(x * S), nb, wave=wave)
    G.configure('bf6')
    out, e1 = res[0], res[1]
    assert not e0 and not e1, (e0 + e1)[:10]
    err = np.abs(from_regs(out) / S + Bsum - ref).max()
    err0 = np.abs(from_regs(out0) / S + Bsum - ref).max()
    print('L_inf after %d blocks: streamed bf6(W) %.3g, converted on chip %.3g; difference between the two %.3g'
          % (nb, err0, err, np.abs(from_regs(out) - from_regs(out0)).max() / S))
    assert err < 2.5e-5 * nb and err < 1.15 * err0 + 1e-6


def test_cxx_packer_matches_python_restatement_f16():
    """R2L_PREC_FP16X3_ASM's stream (r2l_capi.hip pack_body_v3, format 3): 32 KiB chunks of 16 hi + 16 residual fragments of
    256 w, the layer-1 bias carrying the same factor"""
    W = make_weights(2, seed=9, gain=0.5)
    buf, aux_off, tail_off, _ = cxx_pack(*W, e4m3=3)
    img, aux, _ = G.pack_body_image(*W, fmt='f16')
    G.configure('bf6')
    assert aux_off == img.size == 2 * 16 * 32768 and tail_off == aux_off + 2 * G.AUX_BYTES
    assert np.array_equal(buf[:aux_off], img)
    got = buf[aux_off:tail_off].view(np.uint32).reshape(2, -1)
    assert np.array_equal(got[:, :256], aux[:, :256])          # the biases; the scale words are not read by this build
    lo = img.reshape(2, 16, 32, 1024)[:, :, 16:].view(np.float16)
    # a residual is uniform in +-ulp/2 of 256 w: all but the few closest to zero are normal fp16 numbers (unscaled: almost none)
    assert np.abs(lo).max() > 0 and (np.abs(lo[lo != 0]) >= 2.0 ** -14).mean() > 0.9


@pytest.mark.parametrize('nb,wave', [(2, 0), (3, 3)])
def test_f16_stream_three_fp16_passes(nb, wave):
    """R2L_PREC_FP16X3 on the body's machine ('f16'): hi(W) hi(a) + hi(W) lo(a) + lo(W) hi(a) as three fp16 MFMAs per
    k-step on one accumulate chain, lo = the fp16 rounding residual (lo(W) streamed as a second set of fragments, lo(a)
    kept as B operands in AGPRs): no scales, no calibration, fp32-grade result."""
    W = make_weights(nb, seed=30 + nb)
    rng = np.random.default_rng(6)
    x = np.maximum(rng.normal(0, 1, (32, 256)), 0).astype(np.float32)
    S = 16.0
    ref = ref_blocks(x, *W)
    Bsum = np.sum([b.astype(np.float64) for b in W[3]], axis=0)
    img, aux, _ = G.pack_body_image(*W, fmt='f16')
    assert img.size == nb * 16 * 32768
    out, e = G.emulate_tile(G.Opts(fmt='f16'), img, aux, to_regs(x * S), nb, wave=wave)
    G.configure('bf6')
    assert not e, e[:10]
    err = np.abs(from_regs(out) / S + Bsum - ref).max()
    print('L_inf after %d blocks: three fp16 passes %.3g (one pass: %.3g)' % (nb, err, fp16x1_error(x, *W)))
    # the stream holds W x 2^8, so that lo(W) of weights ~ 2^-5 is a normal fp16 number (not one on the 2^-24 subnormal grid)
    assert err < 4e-7 * nb and err < fp16x1_error(x, *W) / 500


@pytest.mark.parametrize('nb,wave,guard', [(2, 0, False), (3, 2, False), (2, 3, True)])
def test_staged_stream_is_bit_identical_to_the_lds_dma_stream(nb, wave, guard):
    """Opts.stage: the weight stream travels through 28 staging AGPRs (global_load_dwordx4, one row tile later
    ds_write_b128 into the ring; the barrier of a rendezvous certifies the stores of the one before) instead of LDS-DMA.
    Same bytes in the same ring slots, same MFMAs: bit-identical results; the emulator checks that no LDS byte is read
    before the barrier that certifies it, that no staging register is stored before its load was waited for and that no
    barrier is entered with a staged store in flight."""
    W = make_weights(nb, seed=40 + nb)
    rng = np.random.default_rng(12)
    x = np.maximum(rng.normal(0, 1, (32, 256)), 0).astype(np.float32)
    img, aux, _ = G.pack_body_image(*W)
    r0 = G.emulate_tile(G.Opts(guard=guard), img, aux, to_regs(x * 16.0), nb, wave=wave)
    r1 = G.emulate_tile(G.Opts(guard=guard, stage=True), img, aux, to_regs(x * 16.0), nb, wave=wave)
    assert not r0[1] and not r1[1], (r0[1][:5], r1[1][:5])
    assert np.array_equal(r0[0], r1[0])
    if guard:
        assert np.array_equal(r0[2], r1[2])


@pytest.mark.parametrize('name,fmt,guard', [('r2l_body_asm.inc', 'bf6', False), ('r2l_body_guard_asm.inc', 'bf6', True),
                                            ('r2l_body8_asm.inc', 'fp8', False), ('r2l_body8_guard_asm.inc', 'fp8', True),
                                            ('r2l_bodyx_asm.inc', 'f16', False)])
def test_committed_asm_is_the_generators_output(tmp_path, name, fmt, guard):
    """The five body streams hipcc assembles are committed text; the emulator tests above run the generator's stream, not that
    text -- this pins one to the other (as tests/test_head_gen_cpu.py and tests/test_nerf_gen_cpu.py do for theirs), with the
    options of csrc/Makefile's rules.  Also what r2l_body.hip's clobber list relies on: the stream saves m0 with its first
    instruction and restores it with its last."""
    out = str(tmp_path / name)
    try:
        G.emit_inc(out, G.Opts(fmt=fmt, guard=guard))
    finally:
        G.configure('bf6')
    built = open(os.path.join(ROOT, 'efficient-nerf_amd', 'csrc', name)).read()
    assert open(out).read() == built, name
    lines = [ln for ln in built.splitlines() if ln.startswith('"')]
    assert lines[0].startswith('"s_mov_b32 s76, m0') and lines[-1].startswith('"s_mov_b32 m0, s76'), (lines[0], lines[-1])
