"""The fp16-only teacher chain as ONE statement with the embedding in the stream (NERF_GEN_FMT=f16c4e -> nerf_mlpx4e_asm.inc,
what R2L_PREC_FP16X1 launches without given view directions), checked on the CPU: the lane-accurate emulator runs the kernel
-- ring prologue, ray loads, the first tile's embedding, tile loop with the next tile's embedding as filler instructions, raw
stores -- on the C++ packer's bytes over several tiles and workgroups, against a float64 evaluation of
run_network (main.py:65-87: pts = rays_o + rays_d z, both embedders, NeRF.forward model/nerf_raybased.py:377-401); the
embedding fragments it leaves in its AGPRs are compared bit for bit with nerf_tile_embed's arithmetic restated in numpy."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'efficient-nerf_amd', 'csrc', 'gen'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_nerf_gen_cpu as T  # noqa: E402
import test_nerf_genx_cpu as TX  # noqa: E402

GE = TX._load_x('f16c4e')


def _scene(n_rays, S, seed, shared_z=False):
    rng = np.random.default_rng(seed)
    ro = (rng.normal(size=(n_rays, 3)) * 0.3 + np.array([0., 0., 4.])).astype(np.float32)
    rd = rng.normal(size=(n_rays, 3)) * 0.25 + np.array([0., 0., -1.])
    rd = (rd * rng.uniform(0.7, 1.3, size=(n_rays, 1))).astype(np.float32)
    z = np.sort(rng.uniform(2., 6., size=(1 if shared_z else n_rays, S)), -1).astype(np.float32)
    return ro, rd, z


def _ref_raw(t, ro, rd, z, S):
    n = ro.shape[0]
    zz = np.broadcast_to(z, (n, S))
    pts = (ro[:, None, :] + rd[:, None, :] * zz[:, :, None]).astype(np.float32).reshape(-1, 3)     # as main.py:701 rounds it
    vd = rd.astype(np.float64) / np.linalg.norm(rd.astype(np.float64), axis=1, keepdims=True)
    vd = np.repeat(vd.astype(np.float32), S, axis=0)
    ref, _, _ = T.ref_mlp(t, pts, vd)
    f16 = T.ref_mlp(t, pts, vd, f16_ops=True)[0]
    return ref, np.abs(f16 - ref).max()


def test_layout_and_committed_text(tmp_path):
    assert GE.EMB and GE.NC == 4 and GE.X1 and GE.N_ANCH == 4976 and GE.STREAM_BYTES == TX.GX.STREAM_BYTES
    assert GE.V_PL < 256 and GE.A_RAW + 32 == 256 and GE.N_SGPR_HI == 96
    t = T.make_tensors(seed=9)
    assert np.array_equal(GE.pack_teacher(t)[0], TX.GX.pack_teacher(t)[0])      # the weight stream is the f16 chain's
    GE.emit_kernel(str(tmp_path), GE.Opts())
    for name in ('nerf_mlpx4e_asm.inc', 'nerf_mlpx4e_clobbers.inc'):
        built = os.path.join(ROOT, 'efficient-nerf_amd', 'csrc', name)
        assert open(os.path.join(str(tmp_path), name)).read() == open(built).read(), name


@pytest.mark.parametrize('d', [1, 2, 3, 5, 7, 64, 100, 192, 255, 256, 1000, 65537, 2 ** 31 - 1])
def test_division_constants(d):
    magic, sh1, sh2 = GE.div_magic(d)
    rng = np.random.default_rng(d)
    ns = np.concatenate([np.arange(0, 70000, 7), rng.integers(0, 2 ** 32, 20000), [2 ** 32 - 1, 2 ** 31, 2 ** 31 - 1],
                         (np.arange(1, 3000) * d - 1) % 2 ** 32, (np.arange(1, 3000) * d) % 2 ** 32]).astype(np.uint64)
    t = (ns * np.uint64(magic)) >> np.uint64(32)
    q = ((t + ((ns - t) >> np.uint64(sh1))) & np.uint64(0xffffffff)) >> np.uint64(sh2)
    assert np.array_equal(q, ns // np.uint64(d))


def test_stream_contents():
    body = GE.block_stream(GE.Opts())
    kinds = {}
    for ins in body:
        kinds[ins.kind] = kinds.get(ins.kind, 0) + 1
    assert kinds['mfma16'] == 4976 and kinds['ds'] == 1326 and kinds['vload'] == 12 and kinds['vstore'] == 4
    assert kinds['trans'] == 4 * (16 + 12)          # per column tile: 16 sines of E, 8 of V, one v_sqrt, three v_rcp
    # every embedding filler sits in front of the last MFMA: nothing of it is exposed
    last = max(i for i, ins in enumerate(body) if ins.kind == 'mfma16')
    assert not any(ins.kind == 'trans' for ins in body[last:])


@pytest.mark.parametrize('wave,block,grid,n_rays,S,shared', [(0, 0, 1, 5, 64, False), (3, 1, 2, 7, 100, False), (1, 0, 2, 3, 192, True)])
def test_emulated_kernel_vs_float64(wave, block, grid, n_rays, S, shared):
    t = T.make_tensors(seed=wave + 3)
    ro, rd, z = _scene(n_rays, S, 40 + wave, shared)
    n_pts = n_rays * S
    ref, f16_err = _ref_raw(t, ro, rd, z, S)
    buf, aux_off = TX.cxx_pack_x(t)
    raw, errs = GE.emulate_kernel(GE.Opts(), buf[:aux_off], buf[aux_off:], ro, rd, z, S, 0 if shared else S, n_pts, wave=wave, block=block, grid=grid)
    assert not errs, errs[:10]
    # the rows this wave of this workgroup owns: points tile * 256 + wave * 64 + [0, 64) of tiles block, block + grid, ...
    mine = np.zeros(n_pts, bool)
    n_tiles = (n_pts + 255) // 256
    for tile in range(block, n_tiles, grid):
        lo = tile * 256 + wave * 64
        mine[lo:min(lo + 64, n_pts)] = True
    assert mine.any() and np.isfinite(raw[mine]).all() and np.isnan(raw[~mine]).all()
    err = np.abs(raw[mine] - ref[mine]).max()
    print('wave %d block %d of %d, %d points: L_inf %.3g (single-pass fp16 operands in float64: %.3g)' % (wave, block, grid, int(mine.sum()), err, f16_err))
    assert err <= 1.5 * f16_err and err < 3e-4 * max(1.0, np.abs(ref).max())


# ---- nerf_tile_embed (csrc/nerf_kernels.hip) restated in numpy, statement for statement, for the bit-for-bit comparison ---------------
F = np.float32


def _fma(a, b, c):
    return (a.astype(np.float64) * np.float64(b) + np.asarray(c, dtype=np.float64)).astype(np.float32)


def _to_rev(x):
    rh = (x * GE.INV2PI_HI).astype(np.float32)
    return rh, _fma(x, GE.INV2PI_LO, _fma(x, GE.INV2PI_HI, -rh))


def _trig_pow2(rh, rl, pw, is_cos):
    t = (rh * F(pw)).astype(np.float32)
    u = (t - np.rint(t)).astype(np.float32)
    g = _fma(rl, F(pw), u)
    arg = np.where(is_cos, (F(0.25) - np.abs(g)).astype(np.float32), g)
    return np.sin(2.0 * np.pi * arg.astype(np.float64)).astype(np.float32)


def _split(a):
    h = a.astype(np.float16)
    return h, (a - h.astype(np.float32)).astype(np.float32).astype(np.float16)


def hip_tile_embed(o, d, z):
    """o, d [16, 3], z [16] of one column tile -> (Eh, El [2][64, 8], Vh, Vl [64, 8]) float16, lane = 16 q + point"""
    lanes = np.arange(64)
    q, n = lanes >> 4, lanes & 15
    o, d, z = o[n], d[n], z[n]
    is_cos = (q & 1) != 0
    nrm = np.sqrt(((d[:, 0] * d[:, 0]).astype(F) + (d[:, 1] * d[:, 1]).astype(F)).astype(F) + (d[:, 2] * d[:, 2]).astype(F)).astype(F)
    xs = [(o[:, k] + (d[:, k] * z).astype(F)).astype(F) for k in range(3)]
    vs = [(d[:, k].astype(np.float64) / nrm.astype(np.float64)).astype(F) for k in range(3)]       # __fdiv_rn: correctly rounded
    r = [_to_rev(x) for x in xs]
    E = [[np.zeros((64, 8), np.float16) for _ in range(2)] for _ in range(2)]
    rr = (np.where(q & 2, r[1][0], r[0][0]), np.where(q & 2, r[1][1], r[0][1]))
    for j in range(8):
        E[0][0][:, j], E[0][1][:, j] = _split((_trig_pow2(rr[0], rr[1], 2.0 ** j, is_cos) * F(16)).astype(F))
    for j in range(8):
        rh = r[0] if j < 2 else (r[1] if j < 4 else r[2])
        a = (np.where(q & 2, rh[0], r[2][0]), np.where(q & 2, rh[1], r[2][1]))
        pq = np.where(q & 2, F(512.0 if j & 1 else 256.0), F(2.0 ** j))
        t = (a[0] * pq).astype(F)
        u = (t - np.rint(t)).astype(F)
        g = (a[1].astype(np.float64) * pq.astype(np.float64) + u.astype(np.float64)).astype(F)
        arg = np.where(is_cos, (F(0.25) - np.abs(g)).astype(F), g)
        val = np.sin(2.0 * np.pi * arg.astype(np.float64)).astype(F)
        if j == 6:
            val = np.where(q & 2, np.where(q == 3, xs[2], xs[0]), val)
        if j == 7:
            val = np.where(q & 2, np.where(q == 3, F(0), xs[1]), val)
        E[1][0][:, j], E[1][1][:, j] = _split((val * F(16)).astype(F))
    rv = _to_rev(np.where(q == 0, vs[0], np.where(q == 1, vs[1], vs[2])))
    Vh, Vl = np.zeros((64, 8), np.float16), np.zeros((64, 8), np.float16)
    for j in range(8):
        t = _trig_pow2(rv[0], rv[1], float(1 << (j & 3)), (j >> 2) != 0)
        idv = vs[j] if j < 3 else np.zeros(64, F)
        Vh[:, j], Vl[:, j] = _split((np.where(q == 3, idv, t) * F(16)).astype(F))
    return E, Vh, Vl


def test_stream_embedding_is_nerf_tile_embeds_bit_for_bit():
    """the filler instructions compute what the HIP prologue of the other builds computes, operation for operation: every one of the
    96 fragment registers of a wave equals the numpy restatement of nerf_tile_embed, including the degenerate lanes (zeros, identity)"""
    rng = np.random.default_rng(3)
    st = GE.NState(2, np.zeros(GE.STREAM_BYTES, np.uint8), np.zeros(GE.AUX_BYTES, np.uint8))
    _, setup = GE.kernel_setup_ops()
    setup(st, dict(npts=10 ** 6, S=64, zs=64, magic=0, sh1=0, sh2=0, tile=0, grid=1, ntiles=1))
    lanes = np.arange(64)
    cols = []
    for c in range(4):
        o = (rng.normal(size=(16, 3)) * 0.5 + np.array([0., 0., 4.])).astype(F)
        d = (rng.normal(size=(16, 3)) * 0.4 + np.array([0., 0., -1.])).astype(F)
        z = rng.uniform(2., 6., size=16).astype(F)
        if c == 1:
            o[0], d[0], z[0] = 0, (0., 0., -1.), 2.0            # a point on an axis: exact zeros and ones in the fragments
        cols.append((o, d, z))
        for k in range(3):
            st.A[GE.RAW_O(c) + k] = o[lanes & 15, k].view(np.uint32)
            st.A[GE.RAW_D(c) + k] = d[lanes & 15, k].view(np.uint32)
        st.A[GE.RAW_Z(c)] = z[lanes & 15].view(np.uint32)
    st.run(GE.first_embed_ops())
    assert not st.errors, st.errors[:5]

    def frag(reg):
        return st.A[reg:reg + 4].T.copy().view(np.float16)          # [64, 8]
    for c, (o, d, z) in enumerate(cols):
        E, Vh, Vl = hip_tile_embed(o, d, z)
        for e in range(2):
            for lo in (0, 1):
                got, want = frag(GE.E_reg('E', e, c, bool(lo))), E[e][lo]
                assert np.array_equal(got.view(np.uint16), want.view(np.uint16)), (c, e, lo, np.argwhere(got.view(np.uint16) != want.view(np.uint16))[:4])
        assert np.array_equal(frag(GE.E_reg('V', 0, c, False)).view(np.uint16), Vh.view(np.uint16)), c
        assert np.array_equal(frag(GE.E_reg('V', 0, c, True)).view(np.uint16), Vl.view(np.uint16)), c
