"""The teacher's layer chain generated WITHOUT its correction terms (NERF_GEN_FMT=f16 -> nerf_mlpx_*.inc, R2L_PREC_FP16X1:
one fp16 pass on the 256-wide sources, hi(W) x (hi(E) + lo(E)) on the embedding k-steps), checked on the CPU like the bf6 chain
(tests/test_nerf_gen_cpu.py): layout constants, the C++ packer against the generator's restatement byte for byte, the
committed text against the generator, and the lane-accurate emulation of the exact instruction stream against a float64
evaluation of NeRF.forward -- its error must be that of single-pass fp16 operands, not more."""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'efficient-nerf_amd', 'csrc', 'gen'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_nerf_gen_cpu as T  # noqa: E402  (tensors, float64 reference, input fragments)

import _pkg  # noqa: E402
_pkg.load()
from efficient_nerf_amd import _lib  # noqa: E402


def _load_x(fmt='f16'):
    """a further instance of the generator module with another format (its tables are built at import)"""
    old = os.environ.get('NERF_GEN_FMT')
    os.environ['NERF_GEN_FMT'] = fmt
    try:
        spec = importlib.util.spec_from_file_location('nerf_gen_' + fmt, os.path.join(ROOT, 'efficient-nerf_amd', 'csrc', 'gen', 'nerf_gen.py'))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
    finally:
        if old is None:
            del os.environ['NERF_GEN_FMT']
        else:
            os.environ['NERF_GEN_FMT'] = old
    return m


GX = _load_x()
G3 = _load_x('f16c3')      # three 16-point column tiles per wave: the same stream, 192-point tiles, activation set Q in AGPRs
G4 = _load_x('f16c4')      # four: 256-point tiles (what R2L_PREC_FP16X1 launches)


def cxx_pack_x(tensors):
    keep, arr = _lib.host_ptrs([torch.from_numpy(np.ascontiguousarray(t)) for t in tensors])
    offs = (C.c_longlong * 1)()
    L = _lib.lib()
    n = L.nerf_debug_pack_chain_host(arr, len(keep), 1, None, 0, offs)
    assert n > 0, L.r2l_last_error()
    buf = np.zeros(n, dtype=np.uint8)
    assert L.nerf_debug_pack_chain_host(arr, len(keep), 1, C.c_void_p(buf.ctypes.data), n, offs) == n
    return buf, int(offs[0])


def test_layout_constants():
    assert GX.X1 and GX.NCH == 44 and GX.NCH % GX.NSLOT == 0 and GX.N_ANCH == 2488 and GX.XPASS == 2 and GX.NT == 154
    assert GX.STREAM_BYTES == 1298432          # NERF_CHAINX_STREAM_BYTES (csrc/nerf_common.h)
    assert T.G.STREAM_BYTES == 2166784 and not T.G.X1      # the other instance is untouched


def test_cxx_packer_matches_python_restatement():
    t = T.make_tensors(seed=5)
    buf, aux_off = cxx_pack_x(t)
    img, aux = GX.pack_teacher(t)
    assert aux_off == img.size == GX.STREAM_BYTES and buf.size == img.size + aux.size
    assert np.array_equal(buf[:aux_off], img)
    assert np.array_equal(buf[aux_off:], aux)
    # any weight range packs: there is no bf6 split to outgrow (the bf6 stream refuses max|w| >= 2^6)
    big = [x * (300.0 if i == 4 else 1.0) for i, x in enumerate(t)]
    assert cxx_pack_x(big)[0].size == buf.size


def test_committed_asm_is_the_generators_output(tmp_path):
    GX.emit(str(tmp_path), GX.Opts())
    for name in ('nerf_mlpx_asm.inc', 'nerf_mlpx_pro_asm.inc', 'nerf_mlpx_clobbers.inc', 'nerf_mlpx_pro_clobbers.inc'):
        built = os.path.join(ROOT, 'efficient-nerf_amd', 'csrc', name)
        assert open(os.path.join(str(tmp_path), name)).read() == open(built).read(), name


@pytest.mark.parametrize('wave,n_tiles,gain', [(0, 1, 1.0), (3, 2, 1.0), (1, 1, 1.5)])
def test_emulated_chain_vs_float64(wave, n_tiles, gain):
    t = T.make_tensors(seed=wave, gain=gain)
    rng = np.random.default_rng(10 + wave)
    pts = rng.uniform(-2.5, 2.5, size=(32, 3)).astype(np.float32)
    vd = rng.normal(size=(32, 3))
    vd = (vd / np.linalg.norm(vd, axis=1, keepdims=True)).astype(np.float32)
    ref, e, v = T.ref_mlp(t, pts, vd)
    f16_err = np.abs(T.ref_mlp(t, pts, vd, f16_ops=True)[0] - ref).max()
    buf, aux_off = cxx_pack_x(t)
    out, errs = GX.emulate_tile(GX.Opts(), buf[:aux_off], buf[aux_off:], T.make_frags(e, v, 16.0), wave=wave, n_tiles=n_tiles)
    assert not errs, errs[:10]
    got = np.zeros((32, 4))
    for c in range(2):
        for k in range(4):
            got[c * 16:(c + 1) * 16, k] = out[c * 4 + k][:16] / 16.0
    err = np.abs(got - ref).max()
    print('wave %d: L_inf %.3g (single-pass fp16 operands in float64: %.3g), |raw| max %.3g' % (wave, err, f16_err, np.abs(ref).max()))
    assert err <= 1.5 * f16_err and err < 3e-4 * max(1.0, np.abs(ref).max())


def test_stream_has_no_correction_terms():
    body = GX.block_stream(GX.Opts())
    kinds = {}
    for ins in body:
        kinds[ins.kind] = kinds.get(ins.kind, 0) + 1
    assert kinds['mfma16'] == 2488 and 'mfma6' not in kinds and kinds['barrier'] == GX.NCH + 1      # 2,632 less the lo(W) pass of the 72 embedding k-step tiles
    assert kinds['dma'] == sum(GX.CHUNKS[(c + 3) % GX.NCH]['pw'] for c in range(GX.NCH))
    assert not any('bf6' in ins.text for ins in body)


# ---- f16c3 / f16c4: three / four column tiles per wave (four is what R2L_PREC_FP16X1 launches) -------------------------------------------------------
def make_frags_nc(G, e, v, act):
    """test_nerf_gen_cpu.make_frags for G.NC column tiles: e [16 NC, 63], v [16 NC, 27] -> {name: uint32 [4, 64]}"""
    lanes = np.arange(64)
    q, n = lanes >> 4, lanes & 15
    fr = {}
    for kind, ne, src in (('E', 2, e), ('V', 1, v)):
        for ee in range(ne):
            for c in range(G.NC):
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


@pytest.mark.parametrize('G', [G3, G4], ids=['three', 'four'])
def test_more_column_tiles_layout_and_committed_text(G, tmp_path):
    nc = G.NC
    assert G.X1 and G.N_ANCH == 1244 * nc and G.NCH == 44 and G.STREAM_BYTES == GX.STREAM_BYTES
    assert len(G.INPUT_NAMES) == 6 * nc and G.N_VGPR_CLOBBER + 4 * nc <= 256 and G.A_E + 24 * nc <= 256
    t = T.make_tensors(seed=9)
    assert np.array_equal(G.pack_teacher(t)[0], GX.pack_teacher(t)[0])          # the weight stream does not know about the tiling
    G.emit(str(tmp_path), G.Opts())
    for name in ('nerf_mlpx%d_asm.inc', 'nerf_mlpx%d_pro_asm.inc', 'nerf_mlpx%d_clobbers.inc', 'nerf_mlpx%d_pro_clobbers.inc'):
        built = os.path.join(ROOT, 'efficient-nerf_amd', 'csrc', name % nc)
        assert open(os.path.join(str(tmp_path), name % nc)).read() == open(built).read(), name % nc


@pytest.mark.parametrize('G,wave,n_tiles,gain', [(G3, 0, 1, 1.0), (G3, 2, 2, 1.5), (G4, 1, 1, 1.0), (G4, 3, 2, 1.5)])
def test_emulated_wider_chain_vs_float64(G, wave, n_tiles, gain):
    G3 = G
    npt = 16 * G.NC
    t = T.make_tensors(seed=wave, gain=gain)
    rng = np.random.default_rng(20 + wave)
    pts = rng.uniform(-2.5, 2.5, size=(npt, 3)).astype(np.float32)
    vd = rng.normal(size=(npt, 3))
    vd = (vd / np.linalg.norm(vd, axis=1, keepdims=True)).astype(np.float32)
    ref, e, v = T.ref_mlp(t, pts, vd)
    f16_err = np.abs(T.ref_mlp(t, pts, vd, f16_ops=True)[0] - ref).max()
    buf, aux_off = cxx_pack_x(t)
    out, errs = G3.emulate_tile(G3.Opts(), buf[:aux_off], buf[aux_off:], make_frags_nc(G3, e, v, 16.0), wave=wave, n_tiles=n_tiles)
    assert not errs, errs[:10]
    got = np.zeros((npt, 4))
    for c in range(G.NC):
        for k in range(4):
            got[c * 16:(c + 1) * 16, k] = out[c * 4 + k][:16] / 16.0
    err = np.abs(got - ref).max()
    print('%d column tiles, wave %d: L_inf %.3g (single-pass fp16 operands in float64: %.3g)' % (G.NC, wave, err, f16_err))
    assert err <= 1.5 * f16_err and err < 3e-4 * max(1.0, np.abs(ref).max())
    kinds = {}
    for ins in G3.block_stream(G3.Opts()):
        kinds[ins.kind] = kinds.get(ins.kind, 0) + 1
    assert kinds['mfma16'] == 1244 * G.NC and 'mfma6' not in kinds and kinds['ds'] == 1326       # the reads of two column tiles feed three / four


# ---- f16p3: fp16x3's arithmetic on the generated chain (R2L_PREC_FP16X3_ASM of the teacher) --------------------------------------------
GP = _load_x('f16p3')


def cxx_pack_fmt(tensors, fmt):
    keep, arr = _lib.host_ptrs([torch.from_numpy(np.ascontiguousarray(t)) for t in tensors])
    offs = (C.c_longlong * 1)()
    L = _lib.lib()
    n = L.nerf_debug_pack_chain_host(arr, len(keep), fmt, None, 0, offs)
    assert n > 0, L.r2l_last_error()
    buf = np.zeros(n, dtype=np.uint8)
    assert L.nerf_debug_pack_chain_host(arr, len(keep), fmt, C.c_void_p(buf.ctypes.data), n, offs) == n
    assert L.nerf_debug_pack_chain_host(arr, len(keep), 7, None, 0, offs) < 0          # the format is an argument, not process state
    return buf, int(offs[0])


def test_three_pass_chain_layout_packer_and_committed_text(tmp_path):
    assert GP.P3 and GP.NC == 2 and GP.NCH == 84 and GP.NCH % GP.NSLOT == 0 and GP.XPASS == 3 and GP.MPASS == 3
    assert GP.N_ANCH == (1100 + 72) * 3 * 2 == 7032       # 1,100 main and 72 embedding k-steps of the chain, three passes, two column tiles
    assert GP.STREAM_BYTES == 2433024                                         # NERF_CHAINP3_STREAM_BYTES (csrc/nerf_common.h)
    for seed, gain in ((5, 1.0), (6, 40.0), (7, 0.01)):                        # the per-layer factor 2^k follows the weights
        t = T.make_tensors(seed=seed, gain=gain)
        buf, aux_off = cxx_pack_fmt(t, 2)
        img, aux = GP.pack_teacher(t)
        assert aux_off == img.size == GP.STREAM_BYTES and buf.size == img.size + aux.size
        assert np.array_equal(buf[:aux_off], img) and np.array_equal(buf[aux_off:], aux)
    GP.emit(str(tmp_path), GP.Opts())
    for name in ('nerf_mlpp3_asm.inc', 'nerf_mlpp3_pro_asm.inc', 'nerf_mlpp3_clobbers.inc', 'nerf_mlpp3_pro_clobbers.inc'):
        built = os.path.join(ROOT, 'efficient-nerf_amd', 'csrc', name)
        assert open(os.path.join(str(tmp_path), name)).read() == open(built).read(), name


@pytest.mark.parametrize('wave,n_tiles,gain', [(0, 1, 1.0), (3, 2, 1.5), (2, 1, 2.5)])
def test_emulated_three_pass_chain_is_fp32_grade(wave, n_tiles, gain):
    """the exact instruction stream on the C++ packer's bytes against float64: two orders of magnitude inside what a single fp16 pass
    gives, and no further from float64 than an fp32 evaluation of the network is"""
    t = T.make_tensors(seed=wave, gain=gain)
    rng = np.random.default_rng(10 + wave)
    pts = rng.uniform(-2.5, 2.5, size=(32, 3)).astype(np.float32)
    vd = rng.normal(size=(32, 3))
    vd = (vd / np.linalg.norm(vd, axis=1, keepdims=True)).astype(np.float32)
    ref, e, v = T.ref_mlp(t, pts, vd)
    f16_err = np.abs(T.ref_mlp(t, pts, vd, f16_ops=True)[0] - ref).max()
    buf, aux_off = cxx_pack_fmt(t, 2)
    out, errs = GP.emulate_tile(GP.Opts(), buf[:aux_off], buf[aux_off:], T.make_frags(e, v, 16.0), wave=wave, n_tiles=n_tiles)
    assert not errs, errs[:10]
    got = np.zeros((32, 4))
    for c in range(2):
        for k in range(4):
            got[c * 16:(c + 1) * 16, k] = out[c * 4 + k][:16] / 16.0
    err = np.abs(got - ref).max()
    print('wave %d gain %g: L_inf %.3g (single-pass fp16 operands: %.3g), |raw| max %.3g' % (wave, gain, err, f16_err, np.abs(ref).max()))
    assert err <= 0.01 * f16_err and err < 1e-6 * max(1.0, np.abs(ref).max())
    kinds = {}
    for ins in GP.block_stream(GP.Opts()):
        kinds[ins.kind] = kinds.get(ins.kind, 0) + 1
    assert kinds['mfma16'] == 7032 and 'mfma6' not in kinds and not any('bf6' in ins.text for ins in GP.block_stream(GP.Opts()))


# ---- mix: the bf6 chain with its first two trunk layers in three passes (R2L_PREC_FP16_MIX: the fine pass of trained teachers, round 6) ----
GM = _load_x('mix')


def test_mixed_chain_layout_packer_and_committed_text(tmp_path):
    assert GM.MIX and GM.MIXK == 2 and not GM.X1 and not GM.P3 and GM.NC == 2 and GM.NCH == 80 and GM.NCH % GM.NSLOT == 0
    assert [l.p3 for l in GM.CHAIN] == [False, True, True] + [False] * 8 and [l.lo_out for l in GM.CHAIN] == [True, True] + [False] * 9
    assert [l.uses_inv for l in GM.CHAIN] == [True, True, True] + [False] * 8 and [l.nj for l in GM.CHAIN] == [0, 0, 0, 4, 4, 4, 4, 4, 4, 4, 2]
    # L1, L2: 16 row tiles x 8 k-steps x 3 passes x 2 column tiles instead of (8 + 4) x 2: 768 more MFMAs per layer than the bf6 chain
    assert GM.N_ANCH == T.G.N_ANCH + 2 * 16 * (8 * 3 * 2 - (8 + 4) * 2) == 4500
    assert GM.STREAM_BYTES == 2232320                                          # NERF_CHAINM_STREAM_BYTES (csrc/nerf_common.h)
    # the lo(a) sets of the three-pass layers: inside the bf6 sets' registers while those are dead, 16 each above the inputs
    regs = {n: sorted({GM.lset(n, s_, c) + i for s_ in range(8) for c in range(2) for i in range(4)}) for n in 'PQ'}
    assert len(regs['P']) == len(regs['Q']) == 64 and not set(regs['P']) & set(regs['Q'])
    assert set(regs['P']) <= set(range(0, 48)) | set(range(240, 256)) and set(regs['Q']) <= set(range(48, 96)) | set(range(224, 240))
    for seed, gain in ((5, 1.0), (6, 40.0), (7, 0.01)):
        t = T.make_tensors(seed=seed, gain=gain)
        buf, aux_off = cxx_pack_fmt(t, 3)
        img, aux = GM.pack_teacher(t)
        assert aux_off == img.size == GM.STREAM_BYTES and buf.size == img.size + aux.size
        assert np.array_equal(buf[:aux_off], img) and np.array_equal(buf[aux_off:], aux)
    GM.emit(str(tmp_path), GM.Opts())
    for name in ('nerf_mlpm_asm.inc', 'nerf_mlpm_pro_asm.inc', 'nerf_mlpm_clobbers.inc', 'nerf_mlpm_pro_clobbers.inc'):
        built = os.path.join(ROOT, 'efficient-nerf_amd', 'csrc', name)
        assert open(os.path.join(str(tmp_path), name)).read() == open(built).read(), name
    assert '"a224"' in open(os.path.join(ROOT, 'efficient-nerf_amd', 'csrc', 'nerf_mlpm_clobbers.inc')).read()        # the block owns a224-a255 too


def _run_tile(G, buf, aux_off, e, v, wave=0, n_tiles=1):
    out, errs = G.emulate_tile(G.Opts(), buf[:aux_off], buf[aux_off:], T.make_frags(e, v, 16.0), wave=wave, n_tiles=n_tiles)
    assert not errs, errs[:10]
    got = np.zeros((32, 4))
    for c in range(2):
        for k in range(4):
            got[c * 16:(c + 1) * 16, k] = out[c * 4 + k][:16] / 16.0
    return got


@pytest.mark.parametrize('wave,n_tiles,gain', [(0, 1, 1.0), (3, 2, 1.5)])
def test_emulated_mixed_chain_on_synthetic_weights(wave, n_tiles, gain):
    """the exact instruction stream (no hazard, every LDS read waited for) on the C++ packer's bytes against float64: on smooth synthetic
    weights the bf6 layers behind the split set the error -- that of the bf6 chain, an order of magnitude inside a single fp16 pass"""
    t = T.make_tensors(seed=wave, gain=gain)
    rng = np.random.default_rng(10 + wave)
    pts = rng.uniform(-2.5, 2.5, size=(32, 3)).astype(np.float32)
    vd = rng.normal(size=(32, 3))
    vd = (vd / np.linalg.norm(vd, axis=1, keepdims=True)).astype(np.float32)
    ref, e, v = T.ref_mlp(t, pts, vd)
    f16_err = np.abs(T.ref_mlp(t, pts, vd, f16_ops=True)[0] - ref).max()
    buf, aux_off = cxx_pack_fmt(t, 3)
    err = np.abs(_run_tile(GM, buf, aux_off, e, v, wave, n_tiles) - ref).max()
    print('wave %d gain %g: mixed chain L_inf %.3g (single-pass fp16 operands: %.3g)' % (wave, gain, err, f16_err))
    assert err <= 0.15 * f16_err
    kinds = {}
    for ins in GM.block_stream(GM.Opts()):
        kinds[ins.kind] = kinds.get(ins.kind, 0) + 1
    assert kinds['mfma16'] == 3656 and kinds['mfma6'] == 844       # the bf6 chain: 2,504 + 1,228


def test_emulated_mixed_chain_on_the_trained_like_fine_network():
    """what the format is for: the committed trained-like FINE network at the 32 densest sample points of four hard rays (sigma up to 31
    there).  Against float64 the bf6 chain's density is 6e-4 (relative) off, the mixed chain's 9e-5, three passes everywhere 5e-6: the
    two exact layers buy a factor of five and more, as the float study on whole frames says (profiles/r06_teacher_mixed_study.txt)"""
    from oracle import r2l_oracle as O
    from oracle import whole_frame as WF
    sds = WF.load_teacher()
    t = [sds[1][n].numpy() for n in O.teacher_state_names()]
    ro, rd = WF.frame_rays(0)
    idx = torch.tensor([44697, 74345, 75007, 45332])
    with torch.no_grad():
        o = O.render_rays_taps(sds[0], sds[1], ro[idx], rd[idx])
    pts, vd = [], []
    for j in range(4):
        z = o['z_vals'][j][torch.topk(o['raw'][j, :, 3], 8).indices]
        pts.append(ro[idx[j]][None] + rd[idx[j]][None] * z[:, None])
        vd.append((rd[idx[j]] / rd[idx[j]].norm())[None].expand(8, 3))
    pts, vd = torch.cat(pts).numpy().astype(np.float32), torch.cat(vd).numpy().astype(np.float32)
    ref, e, v = T.ref_mlp(t, pts, vd)
    assert ref[:, 3].max() > 20.
    rel = {}
    for name, G, fmt in (('bf6', T.G, 0), ('mix', GM, 3), ('p3', GP, 2)):
        buf, aux_off = cxx_pack_fmt(t, fmt)
        d = np.abs(_run_tile(G, buf, aux_off, e, v) - ref)
        rel[name] = (d[:, 3] / np.maximum(1., np.abs(ref[:, 3])))[ref[:, 3] > 1].max()
        print('%-4s raw L_inf rgb %.3g sigma %.3g; sigma relative (sigma > 1) %.3g' % (name, d[:, :3].max(), d[:, 3].max(), rel[name]))
    assert rel['mix'] <= 0.3 * rel['bf6'] and rel['p3'] <= 0.3 * rel['mix'] and rel['mix'] < 2e-4


# ---- f16p3a: the three-pass chain without its view branch (the coarse pass of renders whose caller drops rgb0: nerf_set_skip_rgb0) ----
GA = _load_x('f16p3a')


def test_alpha_only_chain_layout_packer_and_committed_text(tmp_path):
    assert GA.ALPHA and GA.P3 and GA.NC == 2 and len(GA.CHAIN) == 9 and GA.CHAIN[-1].epi == 'alpha' and GA.CHAIN[-1].rt == 2 and GA.CHAIN[-1].fan_out == 1
    assert GA.NCH == 68 and GA.NCH % GA.NSLOT == 0 and GA.STREAM_BYTES == 1998848        # NERF_CHAINP3A_STREAM_BYTES (csrc/nerf_common.h)
    # the view branch's MFMAs are gone: 15 of FA's 17 row tiles, V, RGB
    assert GA.N_ANCH == GP.N_ANCH - (15 * 48 + 8 * (48 + 6) + 24) == 5856
    for seed, gain in ((5, 1.0), (6, 40.0)):
        t = T.make_tensors(seed=seed, gain=gain)
        buf, aux_off = cxx_pack_fmt(t, 4)
        img, aux = GA.pack_teacher(t)
        assert aux_off == img.size == GA.STREAM_BYTES and buf.size == img.size + aux.size
        assert np.array_equal(buf[:aux_off], img) and np.array_equal(buf[aux_off:], aux)
    GA.emit(str(tmp_path), GA.Opts())
    for name in ('nerf_mlpp3a_asm.inc', 'nerf_mlpp3a_pro_asm.inc', 'nerf_mlpp3a_clobbers.inc', 'nerf_mlpp3a_pro_clobbers.inc'):
        built = os.path.join(ROOT, 'efficient-nerf_amd', 'csrc', name)
        assert open(os.path.join(str(tmp_path), name)).read() == open(built).read(), name


@pytest.mark.parametrize('wave,n_tiles,gain', [(0, 1, 1.0), (3, 2, 1.5)])
def test_emulated_alpha_only_chain_gives_the_full_chains_density_bit_for_bit(wave, n_tiles, gain):
    """the density of the chain without its view branch IS the three-pass chain's (same instruction sequence for the trunk and the alpha
    row; its own power-of-two weight scale changes nothing), the colour outputs are zero"""
    t = T.make_tensors(seed=wave, gain=gain)
    rng = np.random.default_rng(10 + wave)
    pts = rng.uniform(-2.5, 2.5, size=(32, 3)).astype(np.float32)
    vd = rng.normal(size=(32, 3))
    vd = (vd / np.linalg.norm(vd, axis=1, keepdims=True)).astype(np.float32)
    ref, e, v = T.ref_mlp(t, pts, vd)
    ba, oa = cxx_pack_fmt(t, 4)
    bp, op = cxx_pack_fmt(t, 2)
    a = _run_tile(GA, ba, oa, e, v, wave, n_tiles)
    p = _run_tile(GP, bp, op, e, v, wave, n_tiles)
    assert np.array_equal(a[:, 3], p[:, 3]) and not a[:, :3].any()
    assert np.abs(a[:, 3] - ref[:, 3]).max() < 1e-6 * max(1.0, np.abs(ref).max())


# ---- f16p3s / mixs: the chains with a second exit behind the density (tiles without a positive density skip the view branch) ----
GS3 = _load_x('f16p3s')
GSM = _load_x('mixs')


@pytest.mark.parametrize('G,fmt,base,stream', [(GS3, 5, 'p3', 2416640), (GSM, 6, 'm', 2220032)])
def test_second_exit_chains_layout_packer_and_committed_text(tmp_path, G, fmt, base, stream):
    assert G.SKIPV and [l.name for l in G.CHAIN] == ['L0', 'L1', 'L2', 'L3', 'L4', 'L5', 'L6', 'L7', 'A', 'F', 'V', 'RGB']
    assert G.CHAIN[8].rt == 1 and G.CHAIN[8].fan_out == 1 and G.CHAIN[9].rt == 16 and G.NCH % G.NSLOT == 0 and G.STREAM_BYTES == stream
    B_ = {'p3': GP, 'm': GM}[base]
    assert G.N_ANCH == B_.N_ANCH and G.NCH == B_.NCH and G.N_SGPR_HI == 60          # the same MFMAs, the alpha row first
    for seed, gain in ((5, 1.0), (6, 40.0)):
        t = T.make_tensors(seed=seed, gain=gain)
        buf, aux_off = cxx_pack_fmt(t, fmt)
        img, aux = G.pack_teacher(t)
        assert aux_off == img.size == G.STREAM_BYTES and buf.size == img.size + aux.size
        assert np.array_equal(buf[:aux_off], img) and np.array_equal(buf[aux_off:], aux)
    G.emit(str(tmp_path), G.Opts())
    for kind in ('asm', 'pro_asm', 'clobbers', 'pro_clobbers'):
        name = 'nerf_mlp%ss_%s.inc' % (base, kind)
        assert open(os.path.join(str(tmp_path), name)).read() == open(os.path.join(ROOT, 'efficient-nerf_amd', 'csrc', name)).read(), name
    text = open(os.path.join(ROOT, 'efficient-nerf_amd', 'csrc', 'nerf_mlp%ss_asm.inc' % base)).read()
    assert text.count('s_cbranch_scc1 L_skip_%=') == 1 and text.count('L_skip_%=:') == 1 and text.count('L_done_%=:') == 1 and '%[fl]' in text
    # every wave passes the same barriers whichever exit the workgroup takes: the cut's own barrier in front of the branch, then either the
    # rest of the full path or the second exit's single one
    body = G.block_stream(G.Opts())
    pre, post = G.split_at_cut(body)
    nb = lambda L: sum(1 for i in L if i.kind == 'barrier')
    assert nb(pre) + nb(post) == nb(B_.block_stream(B_.Opts())) + 1 and nb(G.skip_tail_ops()) == 1
    assert [i.text for i in pre[-3:]] == ['v_readfirstlane_b32 s58, v218', 's_cmp_eq_u32 s58, 0', 's_cbranch_scc1 L_skip_%=']


@pytest.mark.parametrize('G,fmt,B_,bfmt', [(GS3, 5, GP, 2), (GSM, 6, GM, 3)])
def test_emulated_second_exit_is_taken_only_without_a_positive_density_and_changes_no_density(G, fmt, B_, bfmt):
    """two tiles in a row per case (the ring must be primed for the next tile by either exit): with positive densities the full path runs
    and every output is the unsplit chain's bit for bit; with alpha_linear.bias = -100 the second exit is taken, the densities are still
    the unsplit chain's bit for bit and the colours are zero; no hazard, no unwaited LDS read on either path"""
    for wave, shift in ((0, 0.0), (1, -100.0)):
        t = [x.copy() for x in T.make_tensors(seed=wave, gain=1.0)]
        t[21] = t[21] + np.float32(shift)
        rng = np.random.default_rng(10 + wave)
        pts = rng.uniform(-2.5, 2.5, size=(32, 3)).astype(np.float32)
        vd = rng.normal(size=(32, 3))
        vd = (vd / np.linalg.norm(vd, axis=1, keepdims=True)).astype(np.float32)
        ref, e, v = T.ref_mlp(t, pts, vd)
        buf, off = cxx_pack_fmt(t, fmt)
        out, errs, skipped = G.emulate_tile(G.Opts(), buf[:off], buf[off:], T.make_frags(e, v, 16.0), wave=wave, n_tiles=2)
        assert not errs, errs[:10]
        bb, boff = cxx_pack_fmt(t, bfmt)
        want = _run_tile(B_, bb, boff, e, v, wave, 2)
        got = np.zeros((32, 4))
        for c in range(2):
            for k in range(4):
                got[c * 16:(c + 1) * 16, k] = out[c * 4 + k][:16] / 16.0
        assert skipped == [shift < 0] * 2 and np.array_equal(got[:, 3], want[:, 3])
        if shift < 0:
            assert (ref[:, 3] < 0).all() and not got[:, :3].any()
        else:
            assert (ref[:, 3] > 0).any() and np.array_equal(got, want)
