#!/usr/bin/env python3
"""Generator of the hand-scheduled gfx950 head layer of the R2L network for the FP16_FP8 mode:
    h0 = relu(W_h positional_embedding(16 points on the ray) + b_h)        (model/nerf_raybased.py:198-208, 539-541)
Linear(1008, 256) whose input never exists in memory: the 63 embedding values of a point (x, sin / cos(2^l x), l < 10, per
coordinate) are computed in registers while the previous point's MFMAs run, as B operands of the 32x32 shapes
(third instance of the machine of isa.py; the body is body_gen.py).

One straight-line asm block per 128-ray tile (a wave owns 32 rays, lane = 32h + ray); k-outer:
  for point p = 0..15 (one K = 64 group: 63 features + 1 pad):
      embedding of point p+1 (VALU, under this group's MFMAs): x_c = o_c + d_c z_p, then per value
          t = fma(2^l, rh, h/4), u = t - rint(t), sin(2 pi fma(2^l, rl, u))      (rh + rl = x / 2 pi in two floats;
          cos(2 pi g) = sin(2 pi (g + 1/4)); 2^l rh + h/4 is exact for |2^l rh| >= 1/4, v_sin_f32 takes turns)
      -> 16 fp16 pair registers (hi), 16 residual pair registers -> two v_cvt_scalef32_pk32_bf6_f16
      for row tile u = 0..7:  X(u) += hi(W) hi(e)  (4 x v_mfma_f32_32x32x16_f16)
                                     + bf6(W - hi(W)) bf6(e) + bf6(W) bf6(e - hi(e))   (2 x v_mfma_scale_f32_32x32x64_f8f6f4)
  store relu(X) as the register image r2l_body_kernel loads (csrc/r2l_common.h).
X (8 row tiles x 16 AGPRs) starts from the bias; weights and bias are packed x act_scale, so X is in the body's domain.
Element (s, h, j) of the K = 64 group of point p (k-step s of 16, lane half h, element j):
    s < 3:  coordinate s, frequency j, sin (h = 0) | cos (h = 1)
    s = 3:  j < 6: coordinate j >> 1, frequency 8 + (j & 1), sin | cos;  j = 6: x0 | x2;  j = 7: x1 | pad
Weight stream of a tile: 32 chunks of 28 KiB = (point p, row tiles 4m .. 4m+3): 16 fp16 fragments, 8 bf6 operands
(16 B + 8 B per lane); 4-slot LDS ring, LDS-DMA, counted waits: the protocol of body_gen.py / nerf_gen.py with every
address an immediate (a tile starts at stream offset 0; it begins with vmcnt(0) + barrier and ends with the next tile's
first three chunks in flight).

`python head_gen.py --emit DIR` writes r2l_head_asm.inc, r2l_head_pro_asm.inc and their clobber lists;
tests/test_head_gen_cpu.py runs the emulator against a float64 evaluation of the layer.
"""
import argparse
import os
import sys

import numpy as np

import isa
from isa import (Ins, State, Filler, vr, ar, vreg, areg, sreg, mfma32_16, mfma32_6, ds_read_b128, ds_read_b64, waitcnt_lgkm,
                 waitcnt_vm, barrier, valu, v_max0, v_accr, v_cvt_pk_f16, v_resid16, v_cvt_pk32_bf6, s_nop, salu, f_to_bf6,
                 pack6, layer_exponent, weight_exps, f32_bits, check_hazards_stream, model_cycles, v_f32_op, v_fma_f32,
                 v_ldexp_f32, v_rndne_f32, v_sin_f32, v_max3_abs)

# ---------------------------------------------------------------------------------------------
# register map
# ---------------------------------------------------------------------------------------------
V_O, V_D, V_X, V_RH, V_RL = 0, 3, 6, 9, 12
V_BQ = 15          # 0.25 * h: cos(2 pi g) = sin(2 pi (g + 1/4))
V_TT = 16          # 16..27: temporaries of four trig evaluations in flight (3 each)
V_EP = 16          # epilogue: 8 x 4 staging registers (16..47; the embedding is over by then)
V_VAL = 48         # 48..79: the 32 values of a group in f32
V_EH = 80          # 80..111: fp16 hi B operands, 2 buffers x 16
V_E6 = 112         # 112..135: bf6 B operands, 2 buffers x (value 6 | residual 6)
V_LO = 136         # 136..151: fp16 residual pairs
V_HI = 152         # 152..183: fp16 weight fragments, 8 buffers
V_A6 = 184         # 184..195: bf6 weight operands, 2 buffers x 6
V_L0, V_L1, V_L8A, V_L8B, V_AUX, V_LOFF, V_LANE, V_ST = 196, 197, 198, 199, 200, 201, 202, 203
V_SBA, V_SBL, V_CVA, V_CVL = 204, 205, 206, 207
V_SC = 208         # 208, 209: E8M0 weight scales (w - hi | w)
N_VGPR_CLOBBER = 210
V_HMAX = 210       # in/out operand of the block, pinned to v210 by the kernel: running maximum of h0 over the lane's rays
NHI = 8
A_X = 0
N_AGPR_CLOBBER = 128

S_W, S_XOUT, S_WAVE, S_NEG1, S_M0SAVE = 40, 42, 44, 45, 46
S_G = 48           # 48,49 LDS-DMA source
S_WPW = 50         # wave * 7168
S_C = 51           # 51, 52: 1/(2 pi) hi, lo
S_P2 = 54          # 54..63: 2^l, l = 0..9
N_SGPR_LO, N_SGPR_HI = 40, 64

NSLOT = 4
AUX_BYTES = 2048           # 256 f32 bias (x act_scale) | at 1024: (swl, sw, 0, 0) x 4
AUX_SCALES = 1024
NPT = 16                   # points per ray = K = 64 groups
NCH = 2 * NPT              # chunks per tile


def configure(fmt):
    """'bf6': fp16 pass + two bf6 terms (R2L_PREC_FP16_FP8 / _E4M3: the embedding's ranges are fixed, bf6 serves both);
    'f16': three fp16 passes per k-step, hi(W) hi(e) + hi(W) lo(e) + lo(W) hi(e) with lo = the fp16 rounding residual
    (R2L_PREC_FP16X3_ASM: no low-precision term anywhere on the ray's path).  f16 streams lo(W) as a second set of
    fragments (32 KiB chunks: pieces 16 + 4 k + s), keeps lo(e) where bf6 keeps its converted operands, and reads lo(W)
    through a second ring of fragment buffers."""
    global FMT, PIECES, CHUNK, PW, LDS_AUX, LDS_BYTES, STREAM_BYTES, ANCH_PER_RT, ANCH_PER_CHUNK, N_ANCH, NHI, NHL, V_HL, V_EL, NAME
    FMT = fmt
    PIECES = 32 if fmt == 'f16' else 28
    CHUNK = PIECES * 1024
    PW = PIECES // 4
    LDS_AUX = NSLOT * CHUNK
    LDS_BYTES = LDS_AUX + AUX_BYTES
    STREAM_BYTES = NCH * CHUNK
    ANCH_PER_RT = 12 if fmt == 'f16' else 6          # MFMAs of a row tile per point: 4 k-steps x 3 | 4 fp16 + 2 K=64
    ANCH_PER_CHUNK = 4 * ANCH_PER_RT
    N_ANCH = NCH * ANCH_PER_CHUNK
    NHI = 6 if fmt == 'f16' else 8                   # hi(W) fragment buffers at V_HI
    NHL, V_HL = 5, 176                               # f16: lo(W) fragment buffers (176..195)
    V_EL = 112                                       # f16: lo(e) B operands, 2 buffers x 16 (112..143)
    NAME = 'r2l_headx' if fmt == 'f16' else 'r2l_head'


configure('bf6')
EMB_EXP = -1               # embedding values (|sin|, |cos| <= 1, |x| < 14) / 2^-1 fit bf6; residuals 2^-12 finer
RES_SHIFT = 12
INV2PI_HI = np.float32(0.15915494)
INV2PI_LO = np.float32(6.4206382e-09)
J_ORDER = [(0, 0), (1, 0)]  # the two K = 64 MFMAs of a row tile: (term, t)


def X(u):
    return A_X + u * 16


def EH(b, s):
    return V_EH + b * 16 + s * 4


def E6(b, term):
    return V_E6 + b * 12 + term * 6


def EL(b, s):
    return V_EL + b * 16 + s * 4


def head_col(p, s, h, j):
    """column of head.0.weight (reference embedding order per coordinate: sin l = 0..9, cos l = 0..9, x;
    model/nerf_raybased.py:198-208) multiplied by element (s, h, j) of the group of point p; -1 = pad"""
    if s < 3:
        return (3 * p + s) * 21 + (10 if h else 0) + j
    if j < 6:
        return (3 * p + (j >> 1)) * 21 + (10 if h else 0) + 8 + (j & 1)
    if j == 6:
        return (3 * p + (2 if h else 0)) * 21 + 20
    return -1 if h else (3 * p + 1) * 21 + 20


def piece_hi(k, s):
    return k * 4 + s


def piece_a6(k, t):
    return 16 + k * 2 + t


def piece_a6b(k, t):
    return 24 + k, t * 512


def pack_head(W, b, act_scale=16.0, fmt='bf6'):
    """(stream bytes of one tile [STREAM_BYTES], aux bytes [AUX_BYTES]) from head.0.weight [256, 1008], head.0.bias [256]:
    Python restatement of r2l_capi.hip pack_head_v1"""
    configure(fmt)
    Ws = (W.astype(np.float64) * act_scale).astype(np.float32)          # exact: act_scale is a power of two
    hi = Ws.astype(np.float16)
    el, ew = weight_exps(layer_exponent(Ws))
    img = np.zeros(STREAM_BYTES, dtype=np.uint8)
    aux = np.zeros(AUX_BYTES // 4, dtype=np.uint32)
    aux[:256] = (b.astype(np.float64) * act_scale).astype(np.float32).view(np.uint32)
    for q in range(4):
        aux[AUX_SCALES // 4 + 4 * q] = 0x01010101 * (127 + el)
        aux[AUX_SCALES // 4 + 4 * q + 1] = 0x01010101 * (127 + ew)
    lanes = np.arange(64)
    h, r = lanes >> 5, lanes & 31
    for p in range(NPT):
        cols = np.array([[[head_col(p, s, hh, j) for j in range(8)] for s in range(4)] for hh in range(2)])   # [h, s, j]
        for u in range(8):
            base = (2 * p + (u >> 2)) * CHUNK
            k = u & 3
            rows = 32 * u + r
            col = cols[h]                                                   # [64, s, j]
            wv = np.where(col >= 0, Ws[rows[:, None, None], np.maximum(col, 0)], 0).astype(np.float32)      # [64, 4, 8]
            wh = np.where(col >= 0, hi[rows[:, None, None], np.maximum(col, 0)], 0).astype(np.float16)
            for s in range(4):
                o = base + piece_hi(k, s) * 1024
                img[o:o + 1024] = np.ascontiguousarray(wh[:, s, :]).view(np.uint8).reshape(-1)
            if FMT == 'f16':      # pieces 16 + 4 k + s: fp16(w - hi(w))
                wl = (wv.astype(np.float64) - wh.astype(np.float64)).astype(np.float16)
                for s in range(4):
                    o = base + (16 + piece_hi(k, s)) * 1024
                    img[o:o + 1024] = np.ascontiguousarray(wl[:, s, :]).view(np.uint8).reshape(-1)
                continue
            w64, h64 = wv.reshape(64, 32).astype(np.float64), wh.reshape(64, 32).astype(np.float64)
            for t, v in enumerate((np.ldexp(w64 - h64, -el), np.ldexp(w64, -ew))):
                words = pack6(f_to_bf6(v))
                o = base + piece_a6(k, t) * 1024
                img[o:o + 1024] = np.ascontiguousarray(words[:, :4]).view(np.uint8).reshape(-1)
                pc, off = piece_a6b(k, t)
                o = base + pc * 1024 + off
                img[o:o + 512] = np.ascontiguousarray(words[:, 4:]).view(np.uint8).reshape(-1)
    return img, aux.view(np.uint8)


# ---------------------------------------------------------------------------------------------
# builders beyond isa's
# ---------------------------------------------------------------------------------------------
def zop(p):
    return ('s', 'z%d' % p)


def v_sel_half(dst, lo_src, hi_src):
    """dst = lane half 1 ? hi_src : lo_src  (vcc = lanes 32..63, set by the block's setup); a source may be 0.0"""
    def emu(st):
        a = isa._f32(st, lo_src).view(np.uint32)
        b = isa._f32(st, hi_src).view(np.uint32)
        st.V[dst] = np.where(np.arange(64) >= 32, b, a)
    return valu('v_cndmask_b32 %s, %s, %s, vcc' % (vreg(dst), isa._opnd(lo_src), isa._opnd(hi_src)), isa._rd(lo_src, hi_src),
                vr(dst), emu)


def dma_piece(i, tag=''):
    voffr = V_L0 if i < 4 else V_LOFF
    imm = 1024 * (i & 3)
    text = 'global_load_lds_dwordx4 %s, %s offset:%d' % (vreg(voffr), sreg(S_G, 2), imm)

    def emu(st):
        copies = []
        g = st.S[S_G]
        for w in range(4):
            dw = (w - st.wave) * PW * 1024
            for l in range(64):
                src = g + int(st.V[voffr][l]) + dw + imm
                dst = st.m0 + dw + imm + l * 16
                assert 0 <= dst and dst + 16 <= LDS_AUX, dst
                assert 0 <= src and src + 16 <= len(st.img), src
                copies.append((dst, st.img[src:src + 16].copy()))
                st.lds_pending[dst:dst + 16] = True
        st.pend_dma.append(copies)
    return Ins(text, 'dma', rd=vr(voffr), emu=emu, cost=8, tag=tag)


def store_group(i, src):
    """register-image group i (4 f32 per lane) of this wave's x tile <- v[src:src+3]"""
    text = 'global_store_dwordx4 %s, %s, %s offset:%d' % (vreg(V_L0), vreg(src, 4), sreg(S_XOUT, 2), (i & 3) * 1024)

    def emu(st):
        st.xout[i] = st.V[src:src + 4].copy()
    return Ins(text, 'store', rd=vr(V_L0) + vr(src, 4), emu=emu, cost=4)


class HState(State):
    def __init__(self, wave, img, aux, z):
        State.__init__(self, wave, img, np.zeros((1, 1024), dtype=np.uint32), 0, LDS_BYTES)
        self.lds[LDS_AUX:LDS_AUX + len(aux)] = aux
        for p in range(NPT):
            self.S['z%d' % p] = float(z[p])
        self.xout = {}


# ---------------------------------------------------------------------------------------------
# the embedding program of one point
# ---------------------------------------------------------------------------------------------
def embed_ops(p):
    """VALU program that turns (o, d, z_p) into the B operands of group p (buffers p & 1)"""
    b = p & 1
    ops = []
    for c in range(3):                                   # pts = rays_o + rays_d * z (model/nerf_raybased.py:100): two roundings
        ops.append(v_f32_op('mul', V_X + c, zop(p), ('v', V_D + c)))
    for c in range(3):
        ops.append(v_f32_op('add', V_X + c, ('v', V_O + c), ('v', V_X + c)))
    for c in range(3):                                   # x / (2 pi) = rh + rl
        ops.append(v_f32_op('mul', V_RH + c, ('s', S_C), ('v', V_X + c)))
    for c in range(3):
        ops.append(v_fma_f32(V_RL + c, ('v', V_X + c), ('s', S_C), ('v', V_RH + c), neg_c=True))
    for c in range(3):
        ops.append(v_fma_f32(V_RL + c, ('v', V_X + c), ('s', S_C + 1), ('v', V_RL + c)))
    vals = [(8 * c + l, c, l) for c in range(3) for l in range(8)] + [(24 + j, j >> 1, 8 + (j & 1)) for j in range(6)]
    for i0 in range(0, len(vals), 4):
        grp = vals[i0:i0 + 4]
        T = [V_TT + 3 * k for k in range(len(grp))]
        for k, (e, c, l) in enumerate(grp):
            ops.append(v_fma_f32(T[k], ('v', V_RH + c), ('s', S_P2 + l), ('v', V_BQ)))
        for k in range(len(grp)):
            ops.append(v_rndne_f32(T[k] + 1, ('v', T[k])))
        for k in range(len(grp)):
            ops.append(v_f32_op('sub', T[k], ('v', T[k]), ('v', T[k] + 1)))
        for k, (e, c, l) in enumerate(grp):
            ops.append(v_fma_f32(T[k], ('v', V_RL + c), ('s', S_P2 + l), ('v', T[k])))
        for k, (e, c, l) in enumerate(grp):
            ops.append(v_sin_f32(V_VAL + e, ('v', T[k])))
    ops.append(v_sel_half(V_VAL + 30, ('v', V_X + 0), ('v', V_X + 2)))
    ops.append(v_sel_half(V_VAL + 31, ('v', V_X + 1), 0.0))
    for i in range(16):
        ops.append(v_cvt_pk_f16(EH(b, 0) + i, V_VAL + 2 * i, V_VAL + 2 * i + 1))
    for half in range(2):   # half-register writes: all low halves, then all high halves
        for i in range(16):
            ops.append(v_resid16((EL(b, 0) if FMT == 'f16' else V_LO) + i, half, EH(b, 0) + i, half, V_VAL + 2 * i + half, S_NEG1))
    if FMT == 'f16':        # the residual pairs ARE the lo B operands
        return ops
    ops.append(v_cvt_pk32_bf6(E6(b, 0), EH(b, 0), V_CVA))
    ops.append(v_cvt_pk32_bf6(E6(b, 1), V_LO, V_CVL))
    return ops


# ---------------------------------------------------------------------------------------------
# schedule of one tile
# ---------------------------------------------------------------------------------------------
# ANCH_PER_CHUNK (bf6: 24 = 4 row tiles x (4 fp16 + 2 K=64); f16: 48 = 4 x 4 k-steps x 3), N_ANCH: configure()


def anchor(ci, k, kind, sj):
    """anchor number of MFMA (kind, s or t) of the k-th row tile of chunk ci; f16 kinds: 'm16' hi(W) hi(e), 'mhl' hi(W) lo(e),
    'mlh' lo(W) hi(e) of k-step sj, back to back"""
    if FMT == 'f16':
        return ci * ANCH_PER_CHUNK + k * ANCH_PER_RT + 3 * sj + ('m16', 'mhl', 'mlh').index(kind)
    return ci * ANCH_PER_CHUNK + k * 6 + (sj if kind == 'm16' else 4 + sj)


def lds_addr(slot, byte_off, width):
    off = slot * CHUNK + byte_off
    lo, hi = (V_L0, V_L1) if width == 16 else (V_L8A, V_L8B)
    return (lo, off) if off < 65536 else (hi, off - 65536)


def chunk_issue_seq(cn):
    """SALU + LDS-DMA instructions that fetch chunk cn (wrapped into the tile's stream) into its ring slot"""
    off = (cn % NCH) * CHUNK
    slot = cn % NSLOT
    seq = [salu('s_add_u32 %s, %s, %s' % (sreg(S_G), sreg(S_W), sreg(S_WPW)), lambda st: st.S.__setitem__(S_G, st.S[S_W] + st.S[S_WPW])),
           salu('s_addc_u32 %s, %s, 0' % (sreg(S_G + 1), sreg(S_W + 1)))]
    if off:
        seq += [salu('s_add_u32 %s, %s, 0x%x' % (sreg(S_G), sreg(S_G), off), lambda st: st.S.__setitem__(S_G, st.S[S_G] + off)),
                salu('s_addc_u32 %s, %s, 0' % (sreg(S_G + 1), sreg(S_G + 1)))]
    seq.append(salu('s_add_u32 m0, %s, 0x%x' % (sreg(S_WPW), slot * CHUNK), lambda st: setattr(st, 'm0', st.S[S_WPW] + slot * CHUNK)))
    seq.append(s_nop(0))
    for i in range(PW):
        if i == 4:
            seq.append(salu('s_add_u32 m0, m0, 0x1000', lambda st: setattr(st, 'm0', st.m0 + 4096)))
            seq.append(s_nop(0))
        seq.append(dma_piece(i, tag=('dma', cn, i)))
    return seq


class Opts:
    def __init__(self, **kw):
        self.rd_lead = 6
        self.rd_lead6 = 5
        self.cap = 7
        self.dma_gap = 1
        self.pair_waits = True    # one lgkmcnt wait for two consecutive fp16 fragments when both reads are out
        self.fmt = 'bf6'
        self.__dict__.update(kw)


def build_fillers(opts):
    F = []
    # fp16 fragments: one per m16 anchor, NHI rotating buffers
    n16, n6 = 0, 0
    last16, last6 = {}, {}
    HALF = ANCH_PER_CHUNK // 2
    for ci in range(NCH):
        for k in range(4):
            for s in range(4):
                a = anchor(ci, k, 'm16', s)
                prev = last16.get(n16 - NHI, -1)
                pc = piece_hi(k, s)
                bv, off = lds_addr(ci % NSLOT, pc * 1024, 16)
                cert = -1 if ci < 3 else (ci - 1) * ANCH_PER_CHUNK + HALF + 1
                F.append(Filler(ds_read_b128(V_HI + (n16 % NHI) * 4, bv, off, tag=('hi', ci, k, s)),
                                max(prev, a - opts.rd_lead, cert), a, ('rd',)))
                last16[n16] = anchor(ci, k, 'mhl', s) if FMT == 'f16' else a
                if FMT == 'f16':      # lo(W) of the same k-step, second ring
                    al = anchor(ci, k, 'mlh', s)
                    bv, off = lds_addr(ci % NSLOT, (16 + pc) * 1024, 16)
                    F.append(Filler(ds_read_b128(V_HL + (n16 % NHL) * 4, bv, off, tag=('hl', ci, k, s)),
                                    max(last6.get(n16 - NHL, -1), al - opts.rd_lead, cert), al, ('rdl',)))
                    last6[n16] = al
                n16 += 1
            for t in range(2 if FMT != 'f16' else 0):
                a = anchor(ci, k, 'm6', t)
                prev = last6.get(n6 - 2, -1)
                cert = -1 if ci < 3 else (ci - 1) * ANCH_PER_CHUNK + HALF + 1
                e = max(prev, a - opts.rd_lead6, cert)
                bv, off = lds_addr(ci % NSLOT, piece_a6(k, t) * 1024, 16)
                F.append(Filler(ds_read_b128(V_A6 + (n6 & 1) * 6, bv, off, tag=('a6', ci, k, t, 0)), e, a, ('rd6',)))
                pc, po = piece_a6b(k, t)
                bv, off = lds_addr(ci % NSLOT, pc * 1024 + po, 8)
                F.append(Filler(ds_read_b64(V_A6 + (n6 & 1) * 6 + 4, bv, off, tag=('a6', ci, k, t, 1)), e, a, ('rd6',)))
                last6[n6] = a
                n6 += 1
    # embedding of point p + 1 under the MFMAs of point p
    for p in range(NPT - 1):
        a0, a1 = 2 * p * ANCH_PER_CHUNK, (2 * p + 2) * ANCH_PER_CHUNK
        ops = embed_ops(p + 1)
        for i, ins in enumerate(ops):
            F.append(Filler(ins, a0 + (i * (a1 - a0 - 6)) // len(ops), a1 - 3, ('emb',)))
    # rendezvous + refill at the middle of every chunk
    for ci in range(NCH):
        ar = ci * ANCH_PER_CHUNK + HALF
        ch = ('dma',)
        F.append(Filler(waitcnt_vm(PW if ci >= 1 else 0), ar - 1, ar + 1, ch))
        F.append(Filler(barrier(), ar - 1, ar + 1, ch))
        k = 0
        for ins in chunk_issue_seq(ci + 3):
            if ins.kind != 'dma' and k == 0:
                F.append(Filler(ins, ar - 1, ar + 3, ch))
            else:
                F.append(Filler(ins, ar + 1 + opts.dma_gap * k, ar + ANCH_PER_CHUNK - 2, ch))
                if ins.kind == 'dma':
                    k += 1
    return F


class Sched:
    def __init__(self):
        self.out = []
        self.ds_issued = 0
        self.ds_done = 0
        self.ds_index = {}

    def emit(self, ins):
        self.out.append(ins)
        if ins.kind == 'ds':
            self.ds_index[ins.tag] = self.ds_issued
            self.ds_issued += 1

    def need(self, key, also=()):
        """wait for the LDS read `key`; reads in `also` that have been issued ride along (one wait for a run of MFMAs)"""
        idx = self.ds_index[key]
        if idx < self.ds_done:
            return
        for k2 in also:
            if k2 in self.ds_index:
                idx = max(idx, self.ds_index[k2])
        self.emit(waitcnt_lgkm(self.ds_issued - idx - 1))
        self.ds_done = idx + 1


def schedule(opts):
    sch = Sched()
    fillers = build_fillers(opts)
    for i, f in enumerate(fillers):
        f.seq = i
    chains = {}
    for f in fillers:
        chains.setdefault(f.chain, []).append(f)
    for ch in chains.values():
        for i in range(len(ch) - 2, -1, -1):
            ch[i].deadline = min(ch[i].deadline, ch[i + 1].deadline)
    heads = {ch: 0 for ch in chains}

    def ready(pos):
        r = []
        for ch, lst in chains.items():
            i = heads[ch]
            if i < len(lst) and lst[i].earliest <= pos:
                r.append(lst[i])
        r.sort(key=lambda f: (f.deadline, f.seq))
        return r

    def issue(f):
        sch.emit(f.ins)
        heads[f.chain] += 1

    while True:
        r = ready(-1)
        if not r:
            break
        issue(r[0])
    n16 = n6 = 0
    for a in range(N_ANCH):
        while True:
            r = [f for f in ready(a - 1) if f.deadline <= a]
            if not r:
                break
            issue(r[0])
        ci, rem = divmod(a, ANCH_PER_CHUNK)
        k, j = divmod(rem, ANCH_PER_RT)
        p, u = ci >> 1, 4 * (ci & 1) + k
        b = p & 1
        if FMT == 'f16':
            sj, which = divmod(j, 3)
            if which == 0:
                sch.need(('hi', ci, k, sj))
                ins = mfma32_16('a', X(u), V_HI + (n16 % NHI) * 4, EH(b, sj), 'a', X(u), tag=('m16', ci, k, sj))
            elif which == 1:
                ins = mfma32_16('a', X(u), V_HI + (n16 % NHI) * 4, EL(b, sj), 'a', X(u), tag=('mhl', ci, k, sj))
            else:
                sch.need(('hl', ci, k, sj))
                ins = mfma32_16('a', X(u), V_HL + (n16 % NHL) * 4, EH(b, sj), 'a', X(u), tag=('mlh', ci, k, sj))
                n16 += 1
        elif j < 4:
            nxt = [('hi', ci, k, j + 1)] if j < 3 and opts.pair_waits else []
            sch.need(('hi', ci, k, j), nxt)
            ins = mfma32_16('a', X(u), V_HI + (n16 % NHI) * 4, EH(b, j), 'a', X(u), tag=('m16', ci, k, j))
            n16 += 1
        else:
            t = j - 4
            sch.need(('a6', ci, k, t, 1))
            ins = mfma32_6('a', X(u), V_A6 + (n6 & 1) * 6, E6(b, t), V_SC + t, V_SBA if t == 0 else V_SBL,
                           tag=('m6', ci, k, t), bfile='v')
            n6 += 1
        sch.emit(ins)
        budget = opts.cap
        while budget > 0:
            r = ready(a)
            if not r:
                break
            issue(r[0])
            budget -= r[0].ins.cost
    while True:
        r = ready(N_ANCH + 10 ** 6)
        if not r:
            break
        issue(r[0])
    return sch.out


def setup_ops():
    """(text lines, emulator function) of the per-block setup; asm operands: %[wimg] %[wave] %[xout] %[o0..2] %[d0..2] %[z0..15]"""
    L = []
    a = L.append
    a('s_mov_b32 %s, m0' % sreg(S_M0SAVE))
    a('s_mov_b64 %s, %%[wimg]' % sreg(S_W, 2))
    a('s_mov_b64 %s, %%[xout]' % sreg(S_XOUT, 2))
    a('s_mov_b32 %s, %%[wave]' % sreg(S_WAVE))
    a('s_mov_b32 %s, 0xbf800000' % sreg(S_NEG1))
    a('s_mul_i32 %s, %s, 0x%x' % (sreg(S_WPW), sreg(S_WAVE), PW * 1024))
    a('s_mov_b32 %s, 0x%08x' % (sreg(S_C), f32_bits(INV2PI_HI)))
    a('s_mov_b32 %s, 0x%08x' % (sreg(S_C + 1), f32_bits(INV2PI_LO)))
    for l in range(10):
        a('s_mov_b32 %s, 0x%08x' % (sreg(S_P2 + l), f32_bits(2.0 ** l)))
    a('s_mov_b32 vcc_lo, 0')
    a('s_mov_b32 vcc_hi, -1')
    for c in range(3):
        a('v_mov_b32 %s, %%[o%d]' % (vreg(V_O + c), c))
        a('v_mov_b32 %s, %%[d%d]' % (vreg(V_D + c), c))
    a('v_mbcnt_lo_u32_b32 %s, -1, 0' % vreg(V_LANE))
    a('v_mbcnt_hi_u32_b32 %s, -1, %s' % (vreg(V_LANE), vreg(V_LANE)))
    a('v_lshlrev_b32 %s, 4, %s' % (vreg(V_L0), vreg(V_LANE)))
    a('v_add_u32 %s, 0x10000, %s' % (vreg(V_L1), vreg(V_L0)))
    a('v_add_u32 %s, 0x1000, %s' % (vreg(V_LOFF), vreg(V_L0)))
    a('v_lshlrev_b32 %s, 3, %s' % (vreg(V_L8A), vreg(V_LANE)))
    a('v_add_u32 %s, 0x10000, %s' % (vreg(V_L8B), vreg(V_L8A)))
    a('v_lshrrev_b32 %s, 5, %s' % (vreg(V_AUX), vreg(V_LANE)))
    a('v_cvt_f32_u32 %s, %s' % (vreg(V_BQ), vreg(V_AUX)))
    a('v_mul_f32 %s, 0.25, %s' % (vreg(V_BQ), vreg(V_BQ)))
    a('v_lshlrev_b32 %s, 4, %s' % (vreg(V_AUX), vreg(V_AUX)))
    a('v_add_u32 %s, 0x%x, %s' % (vreg(V_AUX), LDS_AUX, vreg(V_AUX)))
    a('v_mov_b32 %s, 0x%08x' % (vreg(V_SBA), 0x01010101 * (127 + EMB_EXP)))
    a('v_mov_b32 %s, 0x%08x' % (vreg(V_SBL), 0x01010101 * (127 + EMB_EXP - RES_SHIFT)))
    a('v_mov_b32 %s, 0x%08x' % (vreg(V_CVA), f32_bits(2.0 ** EMB_EXP)))
    a('v_mov_b32 %s, 0x%08x' % (vreg(V_CVL), f32_bits(2.0 ** (EMB_EXP - RES_SHIFT))))

    def emu(st, o, d):
        lanes = np.arange(64, dtype=np.uint32)
        for c in range(3):
            st.V[V_O + c] = np.asarray(o[c], dtype=np.float32).view(np.uint32)
            st.V[V_D + c] = np.asarray(d[c], dtype=np.float32).view(np.uint32)
        st.V[V_LANE] = lanes
        st.V[V_L0] = lanes * 16
        st.V[V_L1] = lanes * 16 + 65536
        st.V[V_LOFF] = lanes * 16 + 4096
        st.V[V_L8A] = lanes * 8
        st.V[V_L8B] = lanes * 8 + 65536
        st.V[V_BQ] = (0.25 * (lanes >> 5)).astype(np.float32).view(np.uint32)
        st.V[V_AUX] = LDS_AUX + (lanes >> 5) * 16
        st.V[V_SBA] = 0x01010101 * (127 + EMB_EXP)
        st.V[V_SBL] = 0x01010101 * (127 + EMB_EXP - RES_SHIFT)
        st.V[V_CVA] = f32_bits(2.0 ** EMB_EXP)
        st.V[V_CVL] = f32_bits(2.0 ** (EMB_EXP - RES_SHIFT))
        st.S[S_W] = 0
        st.S[S_WPW] = st.wave * PW * 1024
        st.S[S_C] = float(INV2PI_HI)
        st.S[S_C + 1] = float(INV2PI_LO)
        for l in range(10):
            st.S[S_P2 + l] = 2.0 ** l
    return L, emu


def head_ops():
    """exposed start of a tile: ring certified, X <- bias, weight scales, B operands of point 0"""
    ops = [waitcnt_vm(0), barrier()]
    for u in range(8):
        for g in range(4):
            ops.append(ds_read_b128(X(u) + 4 * g, V_AUX, 128 * u + 32 * g, tag=('bias', u, g), dfile='a'))
    if FMT != 'f16':
        ops.append(ds_read_b64(V_SC, V_AUX, AUX_SCALES, tag=('scale',)))
    ops += embed_ops(0)
    ops.append(waitcnt_lgkm(0))
    return ops


def tail_ops():
    """exposed end of a tile: h0 = relu(X) -> the register image"""
    ops = [s_nop(15), s_nop(15)]
    for i in range(32):
        t = V_EP + (i % 8) * 4
        for r in range(4):
            ops.append(v_accr(t + r, X(i >> 2) + 4 * (i & 3) + r))
        for r in range(4):
            ops.append(v_max0(t + r, t + r))
        for r in (0, 2):      # range tracking of every ray (r2l_get_range_status): h0 is the first bf6 operand set of the body
            ops.append(v_max3_abs(V_HMAX, t + r, t + r + 1))
        ops.append(store_group(i, t))
        if i % 4 == 3 and i < 31:
            ops.append(salu('s_add_u32 %s, %s, 0x1000' % (sreg(S_XOUT), sreg(S_XOUT))))
            ops.append(salu('s_addc_u32 %s, %s, 0' % (sreg(S_XOUT + 1), sreg(S_XOUT + 1))))
    ops.append(s_nop(1))      # the last store's data registers are read before anything overwrites them
    return ops


def prologue_ops():
    seq = []
    for k in range(3):
        seq += chunk_issue_seq(k)
    return seq


def block_stream(opts):
    configure(opts.fmt)
    return head_ops() + schedule(opts) + tail_ops()


def emit(dirname, opts):
    configure(opts.fmt)
    setup, _ = setup_ops()
    body = block_stream(opts)
    n = {}
    for ins in body:
        n[ins.kind] = n.get(ins.kind, 0) + 1
    with open(os.path.join(dirname, NAME + '_asm.inc'), 'w') as f:
        f.write('// GENERATED by gen/head_gen.py -- do not edit.  Head layer of one 128-ray tile: %s\n' %
                ', '.join('%s %d' % kv for kv in sorted(n.items())))
        for line in setup + [i.text for i in body] + ['s_mov_b32 m0, %s' % sreg(S_M0SAVE)]:
            f.write('"%s\\n\\t"\n' % line)
    with open(os.path.join(dirname, NAME + '_pro_asm.inc'), 'w') as f:
        f.write('// GENERATED by gen/head_gen.py -- do not edit.  Ring prologue: chunks 0..2 of the stream\n')
        pro = ['s_mov_b32 %s, m0' % sreg(S_M0SAVE), 's_mov_b64 %s, %%[wimg]' % sreg(S_W, 2),
               's_mov_b32 %s, %%[wave]' % sreg(S_WAVE), 's_mul_i32 %s, %s, 0x%x' % (sreg(S_WPW), sreg(S_WAVE), PW * 1024),
               'v_mbcnt_lo_u32_b32 %s, -1, 0' % vreg(V_LANE), 'v_mbcnt_hi_u32_b32 %s, -1, %s' % (vreg(V_LANE), vreg(V_LANE)),
               'v_lshlrev_b32 %s, 4, %s' % (vreg(V_L0), vreg(V_LANE)), 'v_add_u32 %s, 0x1000, %s' % (vreg(V_LOFF), vreg(V_L0))]
        for line in pro + [i.text for i in prologue_ops()] + ['s_mov_b32 m0, %s' % sreg(S_M0SAVE)]:
            f.write('"%s\\n\\t"\n' % line)

    def clob(vregs, na):
        regs = ['v%d' % i for i in vregs] + ['a%d' % i for i in range(na)]
        regs += ['s%d' % i for i in range(N_SGPR_LO, N_SGPR_HI)] + ['vcc', 'scc', 'memory']
        return ', '.join('"%s"' % r for r in regs) + '\n'

    with open(os.path.join(dirname, NAME + '_clobbers.inc'), 'w') as f:
        f.write('// GENERATED by gen/head_gen.py: registers the tile block owns\n' + clob(range(N_VGPR_CLOBBER), N_AGPR_CLOBBER))
    with open(os.path.join(dirname, NAME + '_pro_clobbers.inc'), 'w') as f:
        f.write('// GENERATED by gen/head_gen.py: registers the ring prologue owns\n' + clob([V_L0, V_LOFF, V_LANE], 0))
    return n, body


def emulate_tile(opts, img, aux, o, d, z, wave=0, n_tiles=1, check_hazards=True, body=None):
    """o, d: [3][64] f32 per lane (lane = 32 h + ray), z [16].  Returns (x image uint32 [32, 4, 64] as f32, errors)."""
    configure(opts.fmt)
    body = body or block_stream(opts)
    st = HState(wave, img, aux, z)
    _, setup = setup_ops()
    setup(st, o, d)
    st.run(prologue_ops())
    for _ in range(n_tiles):
        st.xout = {}
        st.run(body)
    errs = list(st.errors)
    if st.pend_ds:
        errs.append('%d LDS reads never waited for' % len(st.pend_ds))
    if check_hazards:
        errs += check_hazards_stream(body)
    out = np.stack([st.xout[i] for i in range(32)]).view(np.float32)
    return out, errs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--emit', help='directory for r2l_head_asm.inc / r2l_head_pro_asm.inc')
    ap.add_argument('--dump')
    ap.add_argument('--rd-lead', type=int, default=6)
    ap.add_argument('--rd-lead6', type=int, default=5)
    ap.add_argument('--cap', type=int, default=7)
    ap.add_argument('--dma-gap', type=int, default=1)
    ap.add_argument('--no-pair-waits', action='store_true')
    ap.add_argument('--fmt', default='bf6', choices=['bf6', 'f16'], help='f16: three fp16 passes (r2l_headx_*.inc, R2L_PREC_FP16X3_ASM)')
    a = ap.parse_args()
    opts = Opts(rd_lead=a.rd_lead, rd_lead6=a.rd_lead6, cap=a.cap, dma_gap=a.dma_gap, pair_waits=not a.no_pair_waits, fmt=a.fmt)
    if a.emit:
        n, body = emit(a.emit, opts)
        print('wrote', a.emit, n, 'model cycles per tile', model_cycles(body))
    if a.dump:
        body = block_stream(opts)
        with open(a.dump, 'w') as f:
            for ins in body:
                f.write(ins.text + '\n')
        print('model cycles per tile', model_cycles(body))


if __name__ == '__main__':
    sys.exit(main())
