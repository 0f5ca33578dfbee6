#!/usr/bin/env python3
"""Generator of the hand-scheduled gfx950 body loop of the R2L W256 ResMLP (fp16 main pass + two bf6 x bf6
correction terms), plus a lane-accurate CPU emulator of the generated stream.

What is computed (reference: model/nerf_raybased.py:443-465, ResMLP.forward, 43 blocks):
    x <- x + W2 relu(W1 x + b1) + b2            (activations in the act_scale domain)
with b2 folded on the host (x~_i = x_i - sum_{j<i} b2_j, b1'_i = b1_i + W1_i sum_{j<i} b2_j), so a
block is  h = relu(W1 x~ + b1'),  x~ += W2 h : the second layer accumulates IN PLACE into the fp32
residual stream, which is the MFMA C/D operand.

Arithmetic of one Linear(256,256):  y = hi(W) hi(a)              v_mfma_f32_16x16x32_f16      (1 pass)
                                      + bf6(W - hi(W)) bf6(a)    v_mfma_scale_f32_16x16x128_f8f6f4, e3m2 x e3m2,
                                      + bf6(W) bf6(a - hi(a))    4x the fp16 rate             (2 x 1/4 pass)
hi = fp16 rounding.  The correction terms need ~3 significant bits; OCP bf6 (e3m2) with one power-of-two scale
per layer and term (E8M0 operand of the instruction) keeps the network at L_inf ~2e-5 for uniform, Laplace,
sparse and outlier-laden weights (tools/quant_study.py); e2m3 weights or fp4 do not.

Machine model (one wave64 = 32 rays = 2 column tiles of 16, 4 waves per workgroup, one per SIMD):
  VGPR   0..63   INh    fp16 B operands of layer 1 (= hi of x~):  c*32 + s*4, s = k-step
        64..127  Hh     fp16 B operands of layer 2 (= hi of h)
       128..143  ACC    layer-1 accumulators, 2 buffers x 2 column tiles x 4
       144..151  BIAS   layer-1 bias of a row tile (C operand of its first MFMA), 2 buffers
       152..167  HI     fp16 weight fragments (A operand), 4 buffers
       168..179  A6     bf6 weight operands (A operand of the K=128 MFMA), 2 buffers x 6
       180..211  LO     fp16 pairs of a - hi(a) of the 8 row tiles being converted, per column tile 16
       212..231  TMP    epilogue temporaries, per column tile 4 values + 6 conversion outputs
       232..     addresses / constants
  AGPR   0..127  X      fp32 residual stream = in-place accumulator of layer 2; X(u,c)+i = feature
                        16u + 4(lane>>4) + i of ray c*16 + (lane&15)
       128..175  IN6    bf6 B operands of layer 1: a (t,c) 6 regs each | a - hi(a)
       176..223  H6     bf6 B operands of layer 2
The weight stream goes global -> LDS ring (4 slots x 28 KiB, LDS-DMA, 3 chunks ahead) -> ds_read.
A chunk = 2 row tiles (32 output features) of one layer: 16 hi fragments (1 KiB) + 8 bf6 operands (1.5 KiB).
One counted vmcnt + one s_barrier per chunk (at its middle).

Every instruction is an `Ins` with its assembly text, the registers it reads / writes and a Python
closure that executes it on the emulator state; `schedule` interleaves the fixed MFMA anchor sequence
with the filler instructions (LDS reads, epilogue VALU, LDS-DMA, waits) by a small list scheduler.
`python body_gen.py --emit r2l_body_asm.inc` writes the inline-asm body; the tests run the emulator
against a float64 reference (tests/test_body_gen_cpu.py).

Hazards the stream must respect by construction (hipcc pads none of them inside an asm statement);
`check_hazards_stream` enforces them statically:
  * MFMA result -> any non-accumulate reader: the reader comes >= 2 further MFMAs later;
  * VALU write -> MFMA operand read: >= 2 instructions in between;
  * a VALU that writes HALF a register (v_fma_mixlo/hi_f16) must not be followed directly by a reader of that
    register ("dst-sel forwarding" hazard: the reader sees the stale half);
  * s_mov m0 -> LDS-DMA: one instruction in between.
"""
import argparse
import sys

import numpy as np

# ---------------------------------------------------------------------------------------------
# register map
# ---------------------------------------------------------------------------------------------
NLANE = 64
V_INH = 0
V_HH = 64
V_ACC = 128
V_BIAS = 144
V_HI = 152
V_A6 = 168
V_LO = 180
V_TMP = 212
V_L0 = 232        # lane*16                (LDS bytes 0 .. 65535)
V_L1 = 233        # lane*16 + 65536
V_L8A = 234       # lane*8                 (8-byte parts of the bf6 operands)
V_L8B = 235       # lane*8 + 65536
V_AUX = 236       # LDS aux base + (lane>>4)*16 (+4096 on odd blocks)
V_DMAOFF = 237    # wave*7168 + lane*16    (pieces 0..3; V_DMAOFF2 = +4096: pieces 4..6)
V_DMAOFF2 = 238
V_AUXOFF = 239    # wave*1024 + lane*16
V_LANE = 240
V_SBA = 241       # E8M0 scale of the bf6 activations
V_SC = 242        # 242,243: E8M0 scales (w - hi | w) of layer 1 of a block; 244,245: of layer 2
V_SBL = 246       # E8M0 scale of the bf6 activation residuals
V_CVA = 247       # f32 divisor of the activation conversion
V_CVL = 248       # f32 divisor of the residual conversion
N_VGPR_USED = 249

A_X = 0
A_IN6 = 128
A_H6 = 176
N_AGPR_USED = 224

# SGPRs owned by the body (clobbered); inputs are copied into them at entry
S_W = 40       # 40,41 weight stream base
S_AUXB = 42    # 42,43 aux base
S_XIN = 44     # 44,45
S_XOUT = 46    # 46,47
S_NTILES = 48
S_NBLOCK = 49
S_TILE = 50
S_GRID = 51
S_WAVE = 52
S_POS = 53     # byte offset of the next chunk to issue
S_END = 54     # n_block * 16 * CHUNK
S_G = 56       # 56,57 global address of the chunk being issued
S_M0SLOT = 58  # 58..61: M0 of slot 0..3 for this wave
S_AUXPOS = 62  # byte offset of the next aux block to issue
S_AUXEND = 63
S_AUXM0 = 64   # M0 of the aux slot to fill next
S_AG = 66      # 66,67 aux global address
S_NEG1 = 68    # -1.0f
S_BLK = 69     # block loop counter
S_T0 = 70      # temporaries 70..75
S_M0SAVE = 76
S_TILEOFF = 78  # 78,79
N_SGPR_LO, N_SGPR_HI = 40, 80

PIECES = 28                # 1 KiB pieces of a chunk
CHUNK = PIECES * 1024      # 28 KiB
PW = PIECES // 4           # LDS-DMA pieces per wave and chunk
NSLOT = 4
LDS_AUX = NSLOT * CHUNK
AUX_BYTES = 4096           # per block: 256 f32 bias | 4 lane quarters x (swl1, sw1, swl2, sw2) | pad
AUX_SCALES = 1024
LDS_BYTES = LDS_AUX + 2 * AUX_BYTES
BF6_TOP = 4                # 28 = 1.75 * 2^4
# activations (act_scale domain, < 2^7) are converted as a / 2^3, their fp16 residuals (< 2^-5) as r * 2^9
ACT_EXP = 3
RES_EXP = -9


def layer_exponent(W):
    """e with max|w| in [2^(e-1), 2^e)"""
    m = float(np.abs(W).max())
    return int(np.frexp(m)[1]) if m > 0 else -4


def weight_exps(e):
    """power-of-two exponents of the two bf6 weight operands of a layer with weight exponent e:
    stored (w - hi(w)) / 2^(e-16) and w / 2^(e-4), both < 2^5 in magnitude"""
    return e - 16, e - 4


def INH(s, c):
    return V_INH + c * 32 + s * 4


def HH(s, c):
    return V_HH + c * 32 + s * 4


def ACC(p, c):
    return V_ACC + p * 8 + c * 4


def BIAS(p):
    return V_BIAS + p * 4


def HI(b):
    return V_HI + b * 4


def A6(b):
    return V_A6 + b * 6


def LO(c):
    return V_LO + c * 16


def TMP(c):
    return V_TMP + c * 10


def X(u, c):
    return A_X + (u * 2 + c) * 4


def B6(base, term, t, c):
    return base + term * 24 + (t * 2 + c) * 6


# order of the four K=128 MFMAs of a row tile: (term, t); term 0 = (w - hi) x bf6(a), 1 = w x bf6(a - hi)
J_ORDER = [(0, 0), (1, 0), (0, 1), (1, 1)]


# ---------------------------------------------------------------------------------------------
# layout maps shared with the host packer (r2l_common.h restated; tests compare both sides)
# ---------------------------------------------------------------------------------------------
def kappa(s, q, j):
    """input feature multiplied by element j of lane quarter q of fp16 k-step s (r2l_kappa)"""
    return 32 * s + 16 * (j >> 2) + 4 * q + (j & 3)


def mix_feat(t, q, e):
    """input feature multiplied by element e (0..31) of lane quarter q of K=128 step t: the conversion takes the
    16 fp16 pair registers of row tiles 8t .. 8t+7 in order, element e = 4 * (row tile & 7) + accumulator register"""
    return 16 * (8 * t + (e >> 2)) + 4 * q + (e & 3)


def piece_hi(upos, s):
    return upos * 8 + s


def piece_a6(upos, j):
    """1 KiB piece with the first 16 B/lane of bf6 operand j of row tile upos"""
    return 16 + upos * 4 + j


def piece_a6b(upos, j):
    """(piece, byte offset inside it) of the last 8 B/lane (64 lanes x 8 B = 512 B) of the operand"""
    return 24 + upos * 2 + (j >> 1), (j & 1) * 512


# ---------------------------------------------------------------------------------------------
# number formats (emulator + python-side packer used by the tests)
# ---------------------------------------------------------------------------------------------
def _bf6_table():
    v = np.zeros(64)
    for b in range(64):
        s, e, m = b >> 5, (b >> 2) & 7, b & 3
        x = (m / 4.0) * 2.0 ** -2 if e == 0 else (1 + m / 4.0) * 2.0 ** (e - 3)
        v[b] = -x if s else x
    return v


BF6 = _bf6_table()
_BF6_POS = BF6[:32]           # ascending


def f_to_bf6(x):
    """nearest e3m2 code (ties to even mantissa), saturating at 28; x float array"""
    x = np.asarray(x, dtype=np.float64)
    a = np.abs(x)
    idx = np.searchsorted(_BF6_POS, a).clip(1, 31)
    lo, hi = _BF6_POS[idx - 1], _BF6_POS[idx]
    up = (a - lo > hi - a) | ((a - lo == hi - a) & (((idx - 1) & 1) == 1))
    code = np.where(up, idx, idx - 1)
    code = np.where(a >= _BF6_POS[31], 31, code)
    return (code | np.where(np.signbit(x), 32, 0)).astype(np.uint8)


def pack6(codes):
    """[..., 32] 6-bit codes -> [..., 6] uint32, element i at bits [6i, 6i+6) (little endian)"""
    codes = np.asarray(codes, dtype=np.uint64)
    out = np.zeros(codes.shape[:-1] + (3,), dtype=np.uint64)
    for i in range(32):
        bit = 6 * i
        w, sh = bit >> 6, bit & 63
        out[..., w] |= codes[..., i] << np.uint64(sh)
        if sh > 58:
            out[..., w + 1] |= codes[..., i] >> np.uint64(64 - sh)
    return np.ascontiguousarray(out).view(np.uint32).reshape(codes.shape[:-1] + (6,))


def unpack6(words):
    """[..., 6] uint32 -> [..., 32] codes"""
    words = np.ascontiguousarray(words, dtype=np.uint32)
    w64 = words.view(np.uint64).reshape(words.shape[:-1] + (3,))
    out = np.zeros(words.shape[:-1] + (32,), dtype=np.uint8)
    for i in range(32):
        bit = 6 * i
        w, sh = bit >> 6, bit & 63
        v = w64[..., w] >> np.uint64(sh)
        if sh > 58:
            v = v | (w64[..., w + 1] << np.uint64(64 - sh))
        out[..., i] = (v & np.uint64(63)).astype(np.uint8)
    return out


def pack_body_image(W1s, b1s, W2s, b2s, act_scale=16.0):
    """Python restatement of the host packer (r2l_capi.hip pack_body_v4): returns (stream bytes,
    aux uint32 [n_block, 1024], total folded bias float64 [256]).  W*: [256, 256] float32 (out, in)."""
    n_block = len(W1s)
    img = np.zeros(n_block * 16 * CHUNK, dtype=np.uint8)
    aux = np.zeros((n_block, AUX_BYTES // 4), dtype=np.uint32)
    Bsum = np.zeros(256, dtype=np.float64)
    lanes = np.arange(64)
    q = lanes >> 4
    r = lanes & 15
    for b in range(n_block):
        b1f = b1s[b].astype(np.float64) + W1s[b].astype(np.float64) @ Bsum
        aux[b, :256] = (b1f * act_scale).astype(np.float32).view(np.uint32)
        for layer, Wl in enumerate((W1s[b], W2s[b])):
            Wl = Wl.astype(np.float32)
            hi = Wl.astype(np.float16)
            ex = layer_exponent(Wl)
            el, ew = weight_exps(ex)
            for qq in range(4):
                aux[b, AUX_SCALES // 4 + 4 * qq + 2 * layer] = 0x01010101 * (127 + el)
                aux[b, AUX_SCALES // 4 + 4 * qq + 2 * layer + 1] = 0x01010101 * (127 + ew)
            for m in range(8):
                base = ((b * 2 + layer) * 8 + m) * CHUNK
                for upos in range(2):
                    u = 2 * m + upos
                    rows = 16 * u + r
                    for s in range(8):
                        p = base + piece_hi(upos, s) * 1024
                        frag = np.zeros((64, 8), dtype=np.float16)
                        for j in range(8):
                            frag[:, j] = hi[rows, kappa(s, q, j)]
                        img[p:p + 1024] = frag.view(np.uint8).reshape(-1)
                    for j, (term, t) in enumerate(J_ORDER):
                        codes = np.zeros((64, 32), dtype=np.uint8)
                        for e in range(32):
                            k = mix_feat(t, q, e)
                            w = Wl[rows, k].astype(np.float64)
                            if term == 0:
                                v = np.ldexp(w - hi[rows, k].astype(np.float64), -el)
                            else:
                                v = np.ldexp(w, -ew)
                            codes[:, e] = f_to_bf6(v)
                        words = pack6(codes)                       # [64, 6]
                        p = base + piece_a6(upos, j) * 1024
                        img[p:p + 1024] = np.ascontiguousarray(words[:, :4]).view(np.uint8).reshape(-1)
                        pc, off = piece_a6b(upos, j)
                        p = base + pc * 1024 + off
                        img[p:p + 512] = np.ascontiguousarray(words[:, 4:]).view(np.uint8).reshape(-1)
        Bsum = Bsum + b2s[b].astype(np.float64)
    return img, aux, Bsum


# ---------------------------------------------------------------------------------------------
# instruction objects
# ---------------------------------------------------------------------------------------------
class Ins:
    __slots__ = ('text', 'kind', 'rd', 'wr', 'emu', 'cost', 'tag', 'partial')

    def __init__(self, text, kind, rd=(), wr=(), emu=None, cost=1, tag='', partial=False):
        self.text = text
        self.kind = kind      # 'mfma16' 'mfma6' 'valu' 'ds' 'dma' 'salu' 'wait' 'barrier' 'nop'
        self.rd = tuple(rd)   # registers read:  ('v', n) / ('a', n)
        self.wr = tuple(wr)
        self.emu = emu
        self.cost = cost      # issue slots (4-cycle units) used by the scheduler's budget
        self.tag = tag
        self.partial = partial  # writes 16 bits of its destination (dst-sel forwarding hazard)


def vr(n, cnt=1):
    return [('v', n + i) for i in range(cnt)]


def ar(n, cnt=1):
    return [('a', n + i) for i in range(cnt)]


def vreg(n, cnt=1):
    return 'v%d' % n if cnt == 1 else 'v[%d:%d]' % (n, n + cnt - 1)


def areg(n, cnt=1):
    return 'a%d' % n if cnt == 1 else 'a[%d:%d]' % (n, n + cnt - 1)


def sreg(n, cnt=1):
    return 's%d' % n if cnt == 1 else 's[%d:%d]' % (n, n + cnt - 1)


# ---- emulator state ---------------------------------------------------------------------------
class State:
    def __init__(self, wave, img, aux, n_block):
        self.V = np.zeros((256, NLANE), dtype=np.uint32)
        self.A = np.zeros((256, NLANE), dtype=np.uint32)
        self.S = {}
        self.lds = np.zeros(LDS_BYTES, dtype=np.uint8)
        self.m0 = 0
        self.wave = wave
        self.img = img                              # uint8 weight stream
        self.aux = aux.view(np.uint8).reshape(-1)   # uint8 view of [n_block, 1024] dwords
        self.n_block = n_block
        self.pend_ds = []                   # [(first reg, data)] in issue order
        self.pend_regs = set()
        self.pend_dma = []                  # [(list of (lds_addr, bytes))] in issue order
        self.cert = None                    # N of the last vmcnt wait
        self.lds_pending = np.zeros(LDS_BYTES, dtype=bool)
        self.n_ins = 0
        self.errors = []

    def regs(self, file):
        return self.V if file == 'v' else self.A

    def f32(self, file, n):
        return self.regs(file)[n].view(np.float32)

    def check_rd(self, ins):
        for r in ins.rd:
            if r in self.pend_regs:
                self.errors.append('ins %d (%s) reads %s%d before its ds_read was waited for' %
                                   (self.n_ins, ins.text, r[0], r[1]))

    def run(self, stream):
        for ins in stream:
            self.check_rd(ins)
            if ins.emu is not None:
                ins.emu(self)
            self.n_ins += 1


# ---- builders ----------------------------------------------------------------------------------
def _halves(regs):
    """[n, 64] uint32 -> [64, 2n] float32 of the packed f16 halves (low half first)"""
    return regs.T.copy().view(np.float16).astype(np.float32)


def mfma16(dfile, d, a, b, cfile, c, tag=''):
    """D[dfile d:d+3] = A(v[a:a+3]) x B(v[b:b+3]) + C[cfile c:c+3]"""
    rf = {'v': vreg, 'a': areg}
    text = 'v_mfma_f32_16x16x32_f16 %s, %s, %s, %s' % (rf[dfile](d, 4), vreg(a, 4), vreg(b, 4), rf[cfile](c, 4))

    def emu(st):
        lanes = np.arange(64)
        Ah = _halves(st.V[a:a + 4])         # [64, 8]
        Bh = _halves(st.V[b:b + 4])
        Am = np.zeros((16, 32))
        Bm = np.zeros((32, 16))
        for j in range(8):
            Am[lanes & 15, 8 * (lanes >> 4) + j] = Ah[:, j]
            Bm[8 * (lanes >> 4) + j, lanes & 15] = Bh[:, j]
        D = Am @ Bm
        C = st.regs(cfile)[c:c + 4].view(np.float32).astype(np.float64)   # [4, 64]
        out = np.zeros((4, 64), dtype=np.float32)
        for i in range(4):
            out[i] = (C[i] + D[4 * (lanes >> 4) + i, lanes & 15]).astype(np.float32)
        st.regs(dfile)[d:d + 4] = out.view(np.uint32)

    rc = vr(c, 4) if cfile == 'v' else ar(c, 4)
    wd = vr(d, 4) if dfile == 'v' else ar(d, 4)
    return Ins(text, 'mfma16', rd=vr(a, 4) + vr(b, 4) + rc, wr=wd, emu=emu, tag=tag)


def mfma6(dfile, d, a, b_agpr, scale_a, scale_b, tag=''):
    """D += A(bf6 v[a:a+5], E8M0 v[scale_a]) x B(bf6 a[b:b+5], E8M0 v[scale_b])"""
    rf = {'v': vreg, 'a': areg}
    text = ('v_mfma_scale_f32_16x16x128_f8f6f4 %s, %s, %s, %s, %s, %s op_sel_hi:[0,0,0] cbsz:3 blgp:3' %
            (rf[dfile](d, 4), vreg(a, 6), areg(b_agpr, 6), rf[dfile](d, 4), vreg(scale_a), vreg(scale_b)))

    def emu(st):
        lanes = np.arange(64)
        Ac = unpack6(st.V[a:a + 6].T.copy())          # [64, 32]
        Bc = unpack6(st.A[b_agpr:b_agpr + 6].T.copy())
        sa = 2.0 ** (int(st.V[scale_a][0] & 0xff) - 127)
        sb = 2.0 ** (int(st.V[scale_b][0] & 0xff) - 127)
        Am = np.zeros((16, 128))
        Bm = np.zeros((128, 16))
        for e in range(32):
            Am[lanes & 15, 32 * (lanes >> 4) + e] = BF6[Ac[:, e]] * sa
            Bm[32 * (lanes >> 4) + e, lanes & 15] = BF6[Bc[:, e]] * sb
        D = Am @ Bm
        C = st.regs(dfile)[d:d + 4].view(np.float32).astype(np.float64)
        out = np.zeros((4, 64), dtype=np.float32)
        for i in range(4):
            out[i] = (C[i] + D[4 * (lanes >> 4) + i, lanes & 15]).astype(np.float32)
        st.regs(dfile)[d:d + 4] = out.view(np.uint32)

    dd = vr(d, 4) if dfile == 'v' else ar(d, 4)
    return Ins(text, 'mfma6', rd=vr(a, 6) + ar(b_agpr, 6) + dd + vr(scale_a) + vr(scale_b), wr=dd, emu=emu, tag=tag)


def _ds_read(width, dst, base_v, off, tag):
    n = width // 4
    assert 0 <= off < 65536 and off % width == 0
    text = 'ds_read_b%d %s, %s offset:%d' % (width * 8, vreg(dst, n), vreg(base_v), off)

    def emu(st):
        addr = st.V[base_v].astype(np.int64) + off
        data = np.zeros((n, 64), dtype=np.uint32)
        for l in range(64):
            a0 = int(addr[l])
            if st.lds_pending[a0:a0 + width].any():
                st.errors.append('ins %d (%s): LDS bytes at %d read before their LDS-DMA was certified' %
                                 (st.n_ins, text, a0))
            data[:, l] = st.lds[a0:a0 + width].view(np.uint32)
        st.pend_ds.append((dst, data))
        st.pend_regs.update(vr(dst, n))

    return Ins(text, 'ds', rd=vr(base_v), wr=vr(dst, n), emu=emu, tag=tag)


def ds_read_b128(dst, base_v, off, tag=''):
    return _ds_read(16, dst, base_v, off, tag)


def ds_read_b64(dst, base_v, off, tag=''):
    return _ds_read(8, dst, base_v, off, tag)


def waitcnt_lgkm(n):
    assert 0 <= n <= 15

    def emu(st):
        while len(st.pend_ds) > n:
            dst, data = st.pend_ds.pop(0)
            st.V[dst:dst + len(data)] = data
            for r in vr(dst, len(data)):
                st.pend_regs.discard(r)
    return Ins('s_waitcnt lgkmcnt(%d)' % n, 'wait', emu=emu)


def waitcnt_vm(n):
    def emu(st):
        st.cert = n
    return Ins('s_waitcnt vmcnt(%d)' % n, 'wait', emu=emu)


def _land_dma(st, keep):
    while len(st.pend_dma) > keep:
        for addr, data in st.pend_dma.pop(0):
            st.lds[addr:addr + len(data)] = data
            st.lds_pending[addr:addr + len(data)] = False


def barrier():
    def emu(st):
        if st.cert is None:
            st.errors.append('ins %d: s_barrier without a preceding vmcnt wait' % st.n_ins)
            return
        _land_dma(st, st.cert)   # every wave waited for all but its `cert` youngest LDS-DMA before arriving
    return Ins('s_barrier', 'barrier', emu=emu)


def valu(text, rd, wr, emu, tag='', partial=False):
    return Ins(text, 'valu', rd=rd, wr=wr, emu=emu, tag=tag, partial=partial)


def v_max0(dst, src):
    def emu(st):
        st.V[dst] = np.maximum(st.f32('v', src), np.float32(0)).view(np.uint32)
    return valu('v_max_f32 %s, 0, %s' % (vreg(dst), vreg(src)), vr(src), vr(dst), emu)


def v_accr(vdst, asrc):
    def emu(st):
        st.V[vdst] = st.A[asrc]
    return valu('v_accvgpr_read_b32 %s, %s' % (vreg(vdst), areg(asrc)), ar(asrc), vr(vdst), emu)


def v_accw(adst, vsrc):
    def emu(st):
        st.A[adst] = st.V[vsrc]
    return valu('v_accvgpr_write_b32 %s, %s' % (areg(adst), vreg(vsrc)), vr(vsrc), ar(adst), emu)


def v_cvt_pk_f16(dst, a, b):
    def emu(st):
        lo = st.f32('v', a).astype(np.float16).view(np.uint16).astype(np.uint32)
        hi = st.f32('v', b).astype(np.float16).view(np.uint16).astype(np.uint32)
        st.V[dst] = lo | (hi << 16)
    return valu('v_cvt_pk_f16_f32 %s, %s, %s' % (vreg(dst), vreg(a), vreg(b)), vr(a) + vr(b), vr(dst), emu)


def v_resid16(dst, dst_high, hpk, half, t):
    """half `dst_high` of dst = fp16(t - (float)half(hpk)):  v_fma_mixlo/hi_f16 dst, hpk.f16[half], -1.0 (SGPR), t"""
    op = 'v_fma_mixhi_f16' if dst_high else 'v_fma_mixlo_f16'
    sel = ' op_sel:[1,0,0]' if half else ''
    text = '%s %s, %s, %s, %s%s op_sel_hi:[1,0,0]' % (op, vreg(dst), vreg(hpk), sreg(S_NEG1), vreg(t), sel)

    def emu(st):
        h = ((st.V[hpk] >> (16 * half)) & 0xffff).astype(np.uint16).view(np.float16).astype(np.float32)
        r = (st.f32('v', t) - h).astype(np.float32).astype(np.float16).view(np.uint16).astype(np.uint32)
        if dst_high:
            st.V[dst] = (st.V[dst] & 0x0000ffff) | (r << 16)
        else:
            st.V[dst] = (st.V[dst] & 0xffff0000) | r
    return valu(text, vr(hpk) + vr(t), vr(dst), emu, partial=True)


def v_cvt_pk32_bf6(dst, src, scale_v):
    """v[dst:dst+5] = bf6(f16 v[src:src+15] / f32 v[scale_v]), 32 elements, element i at bits [6i, 6i+6)"""
    text = 'v_cvt_scalef32_pk32_bf6_f16 %s, %s, %s' % (vreg(dst, 6), vreg(src, 16), vreg(scale_v))

    def emu(st):
        x = _halves(st.V[src:src + 16]).astype(np.float64)          # [64, 32]
        sc = st.f32('v', scale_v).astype(np.float64)[:, None]
        st.V[dst:dst + 6] = pack6(f_to_bf6(x / sc)).T
    return valu(text, vr(src, 16) + vr(scale_v), vr(dst, 6), emu)


def s_nop(n):
    return Ins('s_nop %d' % n, 'nop', cost=n + 1)


def salu(text, emu=None):
    return Ins(text, 'salu', emu=emu)


def dma_piece(i, tag=''):
    """piece i (0..6) of this wave's 7 KiB of a chunk: global_load_lds_dwordx4 v_off, s[S_G:S_G+1] offset:imm
    with LDS destination M0 + imm + lane*16; pieces 4..6 use the +4096 offset register and M0 + 4096."""
    voffr = V_DMAOFF if i < 4 else V_DMAOFF2
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


def dma_aux():
    """global_load_lds_dwordx4 v_auxoff, s[S_AG:S_AG+1]: 1 KiB per wave of the next block's aux block"""
    text = 'global_load_lds_dwordx4 %s, %s' % (vreg(V_AUXOFF), sreg(S_AG, 2))

    def emu(st):
        copies = []
        g = st.S[S_AG]
        for w in range(4):
            dw = (w - st.wave) * 1024
            for l in range(64):
                src = g + int(st.V[V_AUXOFF][l]) + dw
                dst = st.m0 + dw + l * 16
                assert LDS_AUX <= dst and dst + 16 <= LDS_BYTES, dst
                copies.append((dst, st.aux[src:src + 16].copy()))
                st.lds_pending[dst:dst + 16] = True
        st.pend_dma.append(copies)
    return Ins(text, 'dma', rd=vr(V_AUXOFF), emu=emu, cost=8)


# ---------------------------------------------------------------------------------------------
# block schedule
# ---------------------------------------------------------------------------------------------
class Filler:
    __slots__ = ('ins', 'earliest', 'deadline', 'chain', 'seq')

    def __init__(self, ins, earliest, deadline, chain):
        self.ins = ins
        self.earliest = earliest  # may be issued after anchor #earliest has been emitted
        self.deadline = deadline  # must be issued before anchor #deadline
        self.chain = chain        # fillers of one chain keep their order
        self.seq = 0


def slot_of(layer, m):
    return (layer * 8 + m) % NSLOT


def lds_addr(slot, byte_off, width):
    """(base VGPR, immediate offset) of byte `byte_off` of ring slot `slot` for a per-lane width of 16 or 8 bytes"""
    off = slot * CHUNK + byte_off
    lo, hi = (V_L0, V_L1) if width == 16 else (V_L8A, V_L8B)
    return (lo, off) if off < 65536 else (hi, off - 65536)


def tile_anchors(T):
    """the 24 MFMAs of a tile as (kind, s or j, c): fp16 k-steps 0..7, the four K=128 MFMA pairs behind
    k-steps 4..7 (their B operands of the previous layer's last row tiles are converted late)"""
    out = []
    for s in range(8):
        out.append(('m16', s, 0))
        out.append(('m16', s, 1))
        if s >= 4:
            out.append(('m6', s - 4, 0))
            out.append(('m6', s - 4, 1))
    return out


ANCH_PER_TILE = 24


def anchor_index(T, kind, sj, c):
    """global anchor number of an MFMA (T may be <0 or >=32: neighbouring block iterations)"""
    if kind == 'm16':
        k = sj * 2 + c + 2 * max(0, sj - 4)
    else:
        k = (sj + 4) * 2 + 2 + c + 2 * sj
    return T * ANCH_PER_TILE + k


def epilogue_ops(T, c):
    """VALU epilogue of row tile T (block-local tile index, may be -1 = tile 31 of the previous block) for
    column tile c.  Returns [(Ins, consumer)]; consumer: None | ('hi', s) | ('b6', term, t)."""
    Tm = T % 32
    layer, u = Tm >> 4, Tm & 15
    tb = TMP(c)
    t = [tb + i for i in range(4)]
    cv = tb + 4                       # 6 conversion outputs
    lo = LO(c) + 2 * (u & 7)
    ops = []
    if layer == 0:
        for i in range(4):
            ops.append((v_max0(t[i], ACC(T & 1, c) + i), None))
        hset, b6 = HH, A_H6
    else:
        for i in range(4):
            ops.append((v_accr(t[i], X(u, c) + i), None))
        hset, b6 = INH, A_IN6
    h01 = hset(u >> 1, c) + 2 * (u & 1)
    h23 = h01 + 1
    ops.append((v_cvt_pk_f16(h01, t[0], t[1]), ('hi', u >> 1)))
    ops.append((v_cvt_pk_f16(h23, t[2], t[3]), ('hi', u >> 1)))
    # half-register writes: low halves first, then the high halves (never two writers of one register back to back)
    ops.append((v_resid16(lo, 0, h01, 0, t[0]), None))
    ops.append((v_resid16(lo + 1, 0, h23, 0, t[2]), None))
    ops.append((v_resid16(lo, 1, h01, 1, t[1]), None))
    ops.append((v_resid16(lo + 1, 1, h23, 1, t[3]), None))
    if (u & 7) == 7:
        tt = u >> 3
        # 32-wide conversions of the finished group of 8 row tiles; the independent one first (dst-sel forwarding)
        ops.append((v_cvt_pk32_bf6(cv, hset(4 * tt, c), V_CVA), None))
        for i in range(6):
            ops.append((v_accw(B6(b6, 0, tt, c) + i, cv + i), ('b6', 0, tt)))
        ops.append((v_cvt_pk32_bf6(cv, LO(c), V_CVL), None))
        for i in range(6):
            ops.append((v_accw(B6(b6, 1, tt, c) + i, cv + i), ('b6', 1, tt)))
    return ops


class Sched:
    """instruction list of consecutive block iterations; tracks the LDS reads for the counted lgkmcnt waits"""

    def __init__(self, opts):
        self.o = opts
        self.out = []             # (iteration, Ins)
        self.ds_issued = 0        # LDS reads issued so far (global count)
        self.ds_done = 0          # all LDS reads with index < ds_done are known complete
        self.ds_index = {}        # key -> index of its ds_read

    def emit(self, it, ins):
        self.out.append((it, ins))
        if ins.kind == 'ds':
            self.ds_issued += 1

    def need(self, it, key):
        """make sure the LDS read registered under `key` has landed"""
        if key not in self.ds_index and it == 0:
            return  # issued by the iteration before the schedule starts (iteration 0 is never the extracted one)
        idx = self.ds_index[key]
        if idx < self.ds_done:
            return
        self.emit(it, waitcnt_lgkm(self.ds_issued - idx - 1))
        self.ds_done = idx + 1


def build_fillers(it, opts):
    """fillers of block iteration `it` (anchors numbered it*768 + ...)"""
    F = []
    base_anchor = it * 32 * ANCH_PER_TILE

    def A(T, kind, sj, c):
        return base_anchor + anchor_index(T, kind, sj, c)

    for T in range(32):
        layer, u = T >> 4, T & 15
        m, upos = u >> 1, u & 1
        slot = slot_of(layer, m)
        # --- weight operand reads.  hi fragments in pairs of k-steps (buffers: pair p & 1 of 4): issue order
        # hi(2p+1), hi(2p) so that the single counted wait in front of k-step 2p covers the pair; the bf6 operands
        # (2 reads each, buffer (T*4 + j) & 1) are fetched about two k-steps ahead of their MFMA.
        for p in range(4):
            g0 = T * 8 + 2 * p
            gp = g0 - opts.rd_lead
            earliest = A(gp // 8, 'm16', gp % 8, 1)
            deadline = A(T, 'm16', 2 * p, 0)
            grp = []
            if layer == 0 and p == 0:
                grp.append(ds_read_b128(BIAS(T & 1), V_AUX, 64 * u, tag=('bias', it, T)))
            for s_ in (2 * p + 1, 2 * p):
                bv, off = lds_addr(slot, piece_hi(upos, s_) * 1024, 16)
                grp.append(ds_read_b128(HI(s_ & 3), bv, off, tag=('hi', it, T, s_)))
            for ins in grp:
                F.append(Filler(ins, earliest, deadline, ('rd',)))
        for j in range(4):
            gj = T * 4 + j
            prev = gj - 2                                  # last user of the buffer
            earliest = max(A(prev // 4, 'm6', prev % 4, 1), A(T, 'm16', 2 + j, 0) - 1)
            deadline = A(T, 'm6', j, 0)
            bv, off = lds_addr(slot, piece_a6(upos, j) * 1024, 16)
            F.append(Filler(ds_read_b128(A6(gj & 1), bv, off, tag=('a6', it, T, j, 0)), earliest, deadline, ('rd6',)))
            pc, po = piece_a6b(upos, j)
            bv, off = lds_addr(slot, pc * 1024 + po, 8)
            F.append(Filler(ds_read_b64(A6(gj & 1) + 4, bv, off, tag=('a6', it, T, j, 1)), earliest, deadline, ('rd6',)))
        if T == 16:
            # this block's layer-2 scales were read during layer 1; flip to the next block's aux slot, then fetch
            # the next block's layer-1 scales (the running layer 1 is over: its scale registers are free)
            F.append(Filler(valu('v_xor_b32 %s, 0x%x, %s' % (vreg(V_AUX), AUX_BYTES, vreg(V_AUX)), vr(V_AUX), vr(V_AUX),
                                 lambda st: st.V.__setitem__(V_AUX, st.V[V_AUX] ^ AUX_BYTES)),
                            A(T, 'm16', 1, 0), A(T, 'm16', 6, 0), ('auxflip',)))
            F.append(Filler(ds_read_b64(V_SC, V_AUX, AUX_SCALES, tag=('scale', it + 1, 0)),
                            A(T, 'm16', 1, 0), A(T + 1, 'm16', 0, 0), ('auxflip',)))
        if T == 1:
            # layer-2 scales of this block (layer 2 of the previous block is over)
            F.append(Filler(ds_read_b64(V_SC + 2, V_AUX, AUX_SCALES + 8, tag=('scale', it, 1)),
                            A(T, 'm16', 1, 0), A(T + 1, 'm16', 0, 0), ('auxflip',)))
        # --- epilogue of the PREVIOUS tile, under this tile's MFMAs ----------------------------
        Tprev = T - 1
        pu = (Tprev % 32) & 15
        nl_T0 = (Tprev - pu) + 16  # first tile of the consuming layer, block-local
        for c in range(2):
            e0 = A(T, 'm16', 0, 1) + 1 + c  # two further MFMAs behind the last writer of its accumulator
            for ins, cons in epilogue_ops(Tprev, c):
                dl = A(T + 1, 'm16', 0, 0)  # latest: the accumulator buffer is reused by tile T+1
                if cons is not None:
                    if cons[0] == 'hi':
                        first = A(nl_T0, 'm16', cons[1], 0)
                    else:
                        first = A(nl_T0, 'm6', J_ORDER.index((cons[1], cons[2])), 0)
                    dl = min(dl, first - 2)
                F.append(Filler(ins, e0, dl, ('epi', c)))
        # --- rendezvous + refill at the middle of each chunk (start of the upos = 1 tile) -------
        if upos == 1:
            a0 = A(T, 'm16', 0, 0)
            ch = ('dma',)
            F.append(Filler(waitcnt_vm(PW), a0 - 1, a0 + 1, ch))
            F.append(Filler(barrier(), a0 - 1, a0 + 1, ch))
            seq = []
            cidx = layer * 8 + m          # chunk of the block being consumed
            tgt_slot = (cidx + 3) % NSLOT
            seq.append(salu('s_add_u32 %s, %s, %s' % (sreg(S_G), sreg(S_W), sreg(S_POS)),
                            lambda st: st.S.__setitem__(S_G, st.S[S_W] + st.S[S_POS])))
            seq.append(salu('s_addc_u32 %s, %s, 0' % (sreg(S_G + 1), sreg(S_W + 1))))
            seq.append(salu('s_add_u32 %s, %s, 0x%x' % (sreg(S_POS), sreg(S_POS), CHUNK),
                            lambda st: st.S.__setitem__(S_POS, st.S[S_POS] + CHUNK)))
            seq.append(salu('s_cmp_eq_u32 %s, %s' % (sreg(S_POS), sreg(S_END))))
            seq.append(salu('s_cselect_b32 %s, 0, %s' % (sreg(S_POS), sreg(S_POS)),
                            lambda st: st.S.__setitem__(S_POS, 0 if st.S[S_POS] == st.S[S_END] else st.S[S_POS])))
            if cidx == 0:
                # aux block of the NEXT block -> the other aux slot; older than this chunk's pieces
                seq.append(salu('s_add_u32 %s, %s, %s' % (sreg(S_AG), sreg(S_AUXB), sreg(S_AUXPOS)),
                                lambda st: st.S.__setitem__(S_AG, st.S[S_AUXB] + st.S[S_AUXPOS])))
                seq.append(salu('s_addc_u32 %s, %s, 0' % (sreg(S_AG + 1), sreg(S_AUXB + 1))))
                seq.append(salu('s_add_u32 %s, %s, 0x%x' % (sreg(S_AUXPOS), sreg(S_AUXPOS), AUX_BYTES),
                                lambda st: st.S.__setitem__(S_AUXPOS, st.S[S_AUXPOS] + AUX_BYTES)))
                seq.append(salu('s_cmp_eq_u32 %s, %s' % (sreg(S_AUXPOS), sreg(S_AUXEND))))
                seq.append(salu('s_cselect_b32 %s, 0, %s' % (sreg(S_AUXPOS), sreg(S_AUXPOS)),
                                lambda st: st.S.__setitem__(S_AUXPOS, 0 if st.S[S_AUXPOS] == st.S[S_AUXEND] else st.S[S_AUXPOS])))
                seq.append(salu('s_mov_b32 m0, %s' % sreg(S_AUXM0), lambda st: setattr(st, 'm0', st.S[S_AUXM0])))
                seq.append(s_nop(0))
                seq.append(dma_aux())
                seq.append(salu('s_xor_b32 %s, %s, 0x%x' % (sreg(S_AUXM0), sreg(S_AUXM0), AUX_BYTES),
                                lambda st: st.S.__setitem__(S_AUXM0, st.S[S_AUXM0] ^ AUX_BYTES)))
            seq.append(salu('s_mov_b32 m0, %s' % sreg(S_M0SLOT + tgt_slot),
                            lambda st, k=S_M0SLOT + tgt_slot: setattr(st, 'm0', st.S[k])))
            seq.append(s_nop(0))
            for i in range(PW):
                if i == 4:
                    seq.append(salu('s_add_u32 m0, m0, 0x1000', lambda st: setattr(st, 'm0', st.m0 + 4096)))
                    seq.append(s_nop(0))
                seq.append(dma_piece(i, tag=('dma', it, T, i)))
            end = A(T, 'm6', 3, 1)
            if opts.dma_burst:
                for ins in seq:
                    F.append(Filler(ins, a0 - 1, a0 + 1, ch))
            else:
                # SALU prelude right behind the barrier, then the pieces spread over the tile's k-steps
                first_piece = next(i for i, x in enumerate(seq) if x.kind == 'dma' and x.tag)
                for ins in seq[:first_piece]:
                    F.append(Filler(ins, a0 - 1, a0 + 4, ch))
                spots = [A(T, 'm16', s_, 1) for s_ in range(1, 8)]
                k = 0
                for ins in seq[first_piece:]:
                    F.append(Filler(ins, spots[min(k, len(spots) - 1)], end + 1, ch))
                    if ins.kind == 'dma':
                        k += 1
    return F


def schedule(opts, n_iter=3):
    """list-schedule n_iter block iterations; returns [(iteration_of_position, Ins)] where the
    position's iteration is that of the surrounding anchors"""
    sch = Sched(opts)
    fillers = []
    for it in range(n_iter):
        fillers += build_fillers(it, opts)
    for i, f in enumerate(fillers):
        f.seq = i
    chains = {}
    for f in fillers:
        chains.setdefault(f.chain, []).append(f)
    for ch in chains.values():
        ch.sort(key=lambda f: (f.seq,))
        for i in range(len(ch) - 2, -1, -1):   # a filler must not hold up a successor with an earlier deadline
            ch[i].deadline = min(ch[i].deadline, ch[i + 1].deadline)
    heads = {ch: 0 for ch in chains}
    total_anchors = n_iter * 32 * ANCH_PER_TILE

    def ready(pos):
        """chain heads that may issue at anchor position pos, by deadline"""
        r = []
        for ch, lst in chains.items():
            i = heads[ch]
            if i < len(lst) and lst[i].earliest <= pos:
                r.append(lst[i])
        r.sort(key=lambda f: (f.deadline, f.seq))
        return r

    def issue(f, it):
        ins = f.ins
        if ins.kind == 'ds':
            sch.ds_index[ins.tag] = sch.ds_issued
        sch.emit(it, ins)
        heads[f.chain] += 1

    while True:  # pre-region: the prefetch of tile 0 of iteration 0
        r = ready(-1)
        if not r:
            break
        issue(r[0], -1)

    for a in range(total_anchors):
        it = a // (32 * ANCH_PER_TILE)
        T = (a // ANCH_PER_TILE) % 32
        kind, sj, c = tile_anchors(T)[a % ANCH_PER_TILE]
        while True:  # forced fillers: deadline reached
            r = [f for f in ready(a - 1) if f.deadline <= a]
            if not r:
                break
            issue(r[0], it)
        layer, u = T >> 4, T & 15
        hset = INH if layer == 0 else HH
        if a % (32 * ANCH_PER_TILE) == 0:
            # loop head: the prefetch issued by the previous iteration's tail -- or by the prologue, which lacks the
            # tail's other reads, so a counted wait would be too generous there -- is drained completely
            sch.emit(it, waitcnt_lgkm(0))
            sch.ds_done = sch.ds_issued
        if kind == 'm16':
            sch.need(it, ('hi', it, T, sj))
            if layer == 0:
                d = ACC(T & 1, c)
                if sj == 0:
                    sch.need(it, ('bias', it, T))
                    ins = mfma16('v', d, HI(sj & 3), hset(sj, c), 'v', BIAS(T & 1), tag=('m16', it, T, sj, c))
                else:
                    ins = mfma16('v', d, HI(sj & 3), hset(sj, c), 'v', d, tag=('m16', it, T, sj, c))
            else:
                d = X(u, c)
                ins = mfma16('a', d, HI(sj & 3), hset(sj, c), 'a', d, tag=('m16', it, T, sj, c))
            cap = opts.cap16
        else:
            gj = T * 4 + sj
            sch.need(it, ('a6', it, T, sj, 1))
            term, t = J_ORDER[sj]
            if u == 0 and sj == 0 and c == 0:
                sch.need(it, ('scale', it, layer))
            b6 = A_IN6 if layer == 0 else A_H6
            dfile, d = ('v', ACC(T & 1, c)) if layer == 0 else ('a', X(u, c))
            ins = mfma6(dfile, d, A6(gj & 1), B6(b6, term, t, c), V_SC + 2 * layer + term, V_SBA if term == 0 else V_SBL,
                        tag=('m6', it, T, sj, c))
            cap = opts.cap6
        if not (kind == 'm6' and J_ORDER[sj][0] in opts.skip_terms):
            sch.emit(it, ins)
        budget = cap
        while budget > 0:
            r = ready(a)
            if not r:
                break
            issue(r[0], it)
            budget -= r[0].ins.cost
    return sch.out


class Opts:
    def __init__(self, **kw):
        self.rd_lead = 3
        self.cap16 = 3
        self.cap6 = 3
        self.dma_burst = False
        self.skip_terms = ()      # diagnostics: drop the K=128 MFMAs of these correction terms (wrong results)
        self.__dict__.update(kw)


def steady_block(opts):
    """(prologue reads, loop body) -- the body is iteration 1 of a 3-iteration schedule; the prologue is
    the set of LDS reads of iteration 1 that the schedule placed inside iteration 0 (re-issued before
    the loop is entered), in their issue order."""
    out = schedule(opts, 3)
    body = [ins for it, ins in out if it == 1]
    pro = [ins for it, ins in out if it == 0 and ins.kind == 'ds' and ins.tag[1] == 1]
    return pro, body


# ---------------------------------------------------------------------------------------------
# whole-kernel text
# ---------------------------------------------------------------------------------------------
def split_ops(u):
    """standalone split of X row tile u -> the layer-1 operand sets (the layer-2 epilogue without MFMAs)"""
    ops = []
    for c in range(2):
        ops += [ins for ins, _ in epilogue_ops(16 + u, c)]
    return ops


def f32_bits(x):
    return int(np.array([x], dtype=np.float32).view(np.uint32)[0])


def kernel_text(opts):
    """asm text of the whole body kernel (one inline-asm statement).  Inputs (asm operands):
    %0 wimg (s64)  %1 aux (s64)  %2 xin (s64)  %3 xout (s64)  %4 n_tiles  %5 n_block  %6 wave  %7 blockIdx.x
    %8 gridDim.x"""
    pro, body = steady_block(opts)
    L = []
    a = L.append
    a('s_mov_b32 %s, m0' % sreg(S_M0SAVE))
    a('s_mov_b64 %s, %%0' % sreg(S_W, 2))
    a('s_mov_b64 %s, %%1' % sreg(S_AUXB, 2))
    a('s_mov_b64 %s, %%2' % sreg(S_XIN, 2))
    a('s_mov_b64 %s, %%3' % sreg(S_XOUT, 2))
    a('s_mov_b32 %s, %%4' % sreg(S_NTILES))
    a('s_mov_b32 %s, %%5' % sreg(S_NBLOCK))
    a('s_mov_b32 %s, %%6' % sreg(S_WAVE))
    a('s_mov_b32 %s, %%7' % sreg(S_TILE))
    a('s_mov_b32 %s, %%8' % sreg(S_GRID))
    a('s_mov_b32 %s, 0xbf800000' % sreg(S_NEG1))
    # lane id, LDS / DMA offsets
    a('v_mbcnt_lo_u32_b32 %s, -1, 0' % vreg(V_LANE))
    a('v_mbcnt_hi_u32_b32 %s, -1, %s' % (vreg(V_LANE), vreg(V_LANE)))
    a('v_lshlrev_b32 %s, 4, %s' % (vreg(V_L0), vreg(V_LANE)))
    a('v_add_u32 %s, 0x10000, %s' % (vreg(V_L1), vreg(V_L0)))
    a('v_lshlrev_b32 %s, 3, %s' % (vreg(V_L8A), vreg(V_LANE)))
    a('v_add_u32 %s, 0x10000, %s' % (vreg(V_L8B), vreg(V_L8A)))
    a('v_lshrrev_b32 %s, 4, %s' % (vreg(V_AUX), vreg(V_LANE)))
    a('v_lshlrev_b32 %s, 4, %s' % (vreg(V_AUX), vreg(V_AUX)))
    a('v_add_u32 %s, 0x%x, %s' % (vreg(V_AUX), LDS_AUX, vreg(V_AUX)))
    a('s_mul_i32 %s, %s, 0x%x' % (sreg(S_T0), sreg(S_WAVE), PW * 1024))        # wave * 7168
    a('v_add_u32 %s, %s, %s' % (vreg(V_DMAOFF), sreg(S_T0), vreg(V_L0)))
    a('v_add_u32 %s, 0x1000, %s' % (vreg(V_DMAOFF2), vreg(V_DMAOFF)))
    a('s_lshl_b32 %s, %s, 10' % (sreg(S_T0 + 1), sreg(S_WAVE)))                 # wave * 1024
    a('v_add_u32 %s, %s, %s' % (vreg(V_AUXOFF), sreg(S_T0 + 1), vreg(V_L0)))
    for k in range(NSLOT):
        a('s_add_u32 %s, %s, 0x%x' % (sreg(S_M0SLOT + k), sreg(S_T0), k * CHUNK))
    a('s_add_u32 %s, %s, 0x%x' % (sreg(S_AUXM0), sreg(S_T0 + 1), LDS_AUX))
    a('v_mov_b32 %s, 0x%08x' % (vreg(V_SBA), 0x01010101 * (127 + ACT_EXP)))
    a('v_mov_b32 %s, 0x%08x' % (vreg(V_SBL), 0x01010101 * (127 + RES_EXP)))
    a('v_mov_b32 %s, 0x%08x' % (vreg(V_CVA), f32_bits(2.0 ** ACT_EXP)))
    a('v_mov_b32 %s, 0x%08x' % (vreg(V_CVL), f32_bits(2.0 ** RES_EXP)))
    a('s_mul_i32 %s, %s, 0x%x' % (sreg(S_END), sreg(S_NBLOCK), 16 * CHUNK))
    a('s_lshl_b32 %s, %s, 12' % (sreg(S_AUXEND), sreg(S_NBLOCK)))
    a('s_mov_b32 %s, 0' % sreg(S_POS))
    a('s_mov_b32 %s, 0' % sreg(S_AUXPOS))
    a('s_cmp_ge_u32 %s, %s' % (sreg(S_TILE), sreg(S_NTILES)))
    a('s_cbranch_scc1 L_exit_%=')
    # ---- ring prologue: aux(0) -> aux slot 0, chunks 0, 1, 2 -> slots 0, 1, 2 ------------------
    issue_aux = [
        's_add_u32 %s, %s, %s' % (sreg(S_AG), sreg(S_AUXB), sreg(S_AUXPOS)),
        's_addc_u32 %s, %s, 0' % (sreg(S_AG + 1), sreg(S_AUXB + 1)),
        's_add_u32 %s, %s, 0x%x' % (sreg(S_AUXPOS), sreg(S_AUXPOS), AUX_BYTES),
        's_cmp_eq_u32 %s, %s' % (sreg(S_AUXPOS), sreg(S_AUXEND)),
        's_cselect_b32 %s, 0, %s' % (sreg(S_AUXPOS), sreg(S_AUXPOS)),
        's_mov_b32 m0, %s' % sreg(S_AUXM0),
        's_nop 0',
        dma_aux().text,
        's_xor_b32 %s, %s, 0x%x' % (sreg(S_AUXM0), sreg(S_AUXM0), AUX_BYTES),
    ]

    def issue_chunk(slot):
        r = ['s_add_u32 %s, %s, %s' % (sreg(S_G), sreg(S_W), sreg(S_POS)),
             's_addc_u32 %s, %s, 0' % (sreg(S_G + 1), sreg(S_W + 1)),
             's_add_u32 %s, %s, 0x%x' % (sreg(S_POS), sreg(S_POS), CHUNK),
             's_cmp_eq_u32 %s, %s' % (sreg(S_POS), sreg(S_END)),
             's_cselect_b32 %s, 0, %s' % (sreg(S_POS), sreg(S_POS)),
             's_mov_b32 m0, %s' % sreg(S_M0SLOT + slot),
             's_nop 0']
        for i in range(PW):
            if i == 4:
                r += ['s_add_u32 m0, m0, 0x1000', 's_nop 0']
            r.append(dma_piece(i).text)
        return r

    L += issue_aux
    for k in range(3):
        L += issue_chunk(k)
    a('s_waitcnt vmcnt(%d)' % (2 * PW))
    a('s_barrier')
    # ---- tile loop ---------------------------------------------------------------------------
    a('L_tile_%=:')
    # x tile address: xin + tile*131072 + wave*32768 + lane*16  (register image [u*2+c][lane][4])
    a('s_lshl_b32 %s, %s, 17' % (sreg(S_TILEOFF), sreg(S_TILE)))
    a('s_lshr_b32 %s, %s, 15' % (sreg(S_TILEOFF + 1), sreg(S_TILE)))
    a('s_lshl_b32 %s, %s, 15' % (sreg(S_T0 + 2), sreg(S_WAVE)))
    a('s_add_u32 %s, %s, %s' % (sreg(S_TILEOFF), sreg(S_TILEOFF), sreg(S_T0 + 2)))
    a('s_addc_u32 %s, %s, 0' % (sreg(S_TILEOFF + 1), sreg(S_TILEOFF + 1)))
    a('s_add_u32 %s, %s, %s' % (sreg(S_T0 + 4), sreg(S_XIN), sreg(S_TILEOFF)))
    a('s_addc_u32 %s, %s, %s' % (sreg(S_T0 + 5), sreg(S_XIN + 1), sreg(S_TILEOFF + 1)))
    for i in range(32):
        a('global_load_dwordx4 %s, %s, %s offset:%d' % (areg(A_X + 4 * i, 4), vreg(V_L0), sreg(S_T0 + 4, 2), (i % 4) * 1024))
        if i % 4 == 3:
            a('s_add_u32 %s, %s, 0x1000' % (sreg(S_T0 + 4), sreg(S_T0 + 4)))
            a('s_addc_u32 %s, %s, 0' % (sreg(S_T0 + 5), sreg(S_T0 + 5)))
    a('s_waitcnt vmcnt(0)')
    # initial split of row tiles 0..14 (tile 15's runs at the head of the loop body)
    for u in range(15):
        for ins in split_ops(u):
            a(ins.text)
    for ins in pro:   # LDS reads the loop head expects in flight
        a(ins.text)
    a('s_mov_b32 %s, %s' % (sreg(S_BLK), sreg(S_NBLOCK)))
    a('L_block_%=:')
    for ins in body:
        a(ins.text)
    a('s_sub_u32 %s, %s, 1' % (sreg(S_BLK), sreg(S_BLK)))
    a('s_cmp_lg_u32 %s, 0' % sreg(S_BLK))
    a('s_cbranch_scc1 L_block_%=')
    # drain the prefetch reads, let the last MFMAs retire, store x
    a('s_waitcnt lgkmcnt(0)')
    a('s_nop 15')
    a('s_nop 15')
    a('s_add_u32 %s, %s, %s' % (sreg(S_T0 + 4), sreg(S_XOUT), sreg(S_TILEOFF)))
    a('s_addc_u32 %s, %s, %s' % (sreg(S_T0 + 5), sreg(S_XOUT + 1), sreg(S_TILEOFF + 1)))
    for i in range(32):
        a('global_store_dwordx4 %s, %s, %s offset:%d' % (vreg(V_L0), areg(A_X + 4 * i, 4), sreg(S_T0 + 4, 2), (i % 4) * 1024))
        if i % 4 == 3:
            a('s_add_u32 %s, %s, 0x1000' % (sreg(S_T0 + 4), sreg(S_T0 + 4)))
            a('s_addc_u32 %s, %s, 0' % (sreg(S_T0 + 5), sreg(S_T0 + 5)))
    a('s_add_u32 %s, %s, %s' % (sreg(S_TILE), sreg(S_TILE), sreg(S_GRID)))
    a('s_cmp_lt_u32 %s, %s' % (sreg(S_TILE), sreg(S_NTILES)))
    a('s_cbranch_scc1 L_tile_%=')
    a('L_exit_%=:')
    a('s_waitcnt vmcnt(0) lgkmcnt(0)')
    a('s_barrier')
    a('s_mov_b32 m0, %s' % sreg(S_M0SAVE))
    return L, pro, body


def emit_inc(path, opts):
    L, pro, body = kernel_text(opts)
    drop = getattr(opts, 'drop', ())   # diagnostics only (wrong results): timing knock-outs of instruction classes
    if drop:
        texts = set()
        for ins in body:
            gone = ins.kind in drop or (ins.kind == 'wait' and 'lgkm' in drop and 'lgkmcnt' in ins.text)
            if gone:
                texts.add(ins.text)
        keep_always = ('s_waitcnt vmcnt', 's_barrier')
        first = L.index('L_block_%=:')
        last = len(L) - 1 - L[::-1].index('s_sub_u32 %s, %s, 1' % (sreg(S_BLK), sreg(S_BLK)))
        L = L[:first + 1] + [t for t in L[first + 1:last] if not (t in texts and not t.startswith(keep_always))] + L[last:]
    if getattr(opts, 'sim32', False):
        # timing model only (wrong results): every pair of 16x16 MFMAs (column tiles 0, 1) becomes ONE 32x32 MFMA of the
        # same FLOPs and the same operand registers; everything else stays
        import re
        first = L.index('L_block_%=:')
        out = []
        for t in L[first:]:
            m = re.match(r'v_mfma_f32_16x16x32_f16 ([va])\[(\d+):(\d+)\], (v\[\d+:\d+\]), v\[(\d+):(\d+)\], ', t)
            m6 = re.match(r'v_mfma_scale_f32_16x16x128_f8f6f4 ([va])\[(\d+):(\d+)\], (\S+), a\[(\d+):(\d+)\], \S+ (v\d+), (v\d+) (.*)', t)
            if m:
                f_, d0, a_, b0 = m.group(1), int(m.group(2)), m.group(4), int(m.group(5))
                if (b0 % 64) >= 32:      # column tile 1: folded into its partner
                    continue
                d = (d0 // 16) * 16
                out.append('v_mfma_f32_32x32x16_f16 %s[%d:%d], %s, v[%d:%d], %s[%d:%d]' % (f_, d, d + 15, a_, b0, b0 + 3, f_, d, d + 15))
            elif m6:
                f_, d0, a_, b0 = m6.group(1), int(m6.group(2)), m6.group(4).rstrip(','), int(m6.group(5))
                if ((b0 - 128) // 6) % 2 == 1:
                    continue
                d = (d0 // 16) * 16
                out.append('v_mfma_scale_f32_32x32x64_f8f6f4 %s[%d:%d], %s, a[%d:%d], %s[%d:%d], %s, %s %s' %
                           (f_, d, d + 15, a_, b0, b0 + 5, f_, d, d + 15, m6.group(7), m6.group(8), m6.group(9)))
            else:
                out.append(t)
        L = L[:first] + out
    n = {}
    for ins in body:
        n[ins.kind] = n.get(ins.kind, 0) + 1
    with open(path, 'w') as f:
        f.write('// GENERATED by gen/body_gen.py -- do not edit.  Loop body (one ResMLP block): %s\n' %
                ', '.join('%s %d' % kv for kv in sorted(n.items())))
        for line in L:
            f.write('"%s\\n\\t"\n' % line)
    return n


# ---------------------------------------------------------------------------------------------
# emulation of one wave over one tile (tests)
# ---------------------------------------------------------------------------------------------
def emulate_tile(opts, img, aux, x_tile_regs, n_block, wave=0, check_hazards=True):
    """x_tile_regs: float32 [128, 64] register image of one wave's X.  Returns (X out [128, 64], errors)."""
    pro, body = steady_block(opts)
    st = State(wave, img, aux, n_block)
    lanes = np.arange(64, dtype=np.uint32)
    st.V[V_LANE] = lanes
    st.V[V_L0] = lanes * 16
    st.V[V_L1] = lanes * 16 + 65536
    st.V[V_L8A] = lanes * 8
    st.V[V_L8B] = lanes * 8 + 65536
    st.V[V_AUX] = LDS_AUX + (lanes >> 4) * 16
    st.V[V_DMAOFF] = wave * PW * 1024 + lanes * 16
    st.V[V_DMAOFF2] = wave * PW * 1024 + lanes * 16 + 4096
    st.V[V_AUXOFF] = wave * 1024 + lanes * 16
    st.V[V_SBA] = 0x01010101 * (127 + ACT_EXP)
    st.V[V_SBL] = 0x01010101 * (127 + RES_EXP)
    st.V[V_CVA] = f32_bits(2.0 ** ACT_EXP)
    st.V[V_CVL] = f32_bits(2.0 ** RES_EXP)
    S = st.S
    S[S_W] = 0
    S[S_AUXB] = 0
    S[S_POS] = 0
    S[S_AUXPOS] = 0
    S[S_END] = n_block * 16 * CHUNK
    S[S_AUXEND] = n_block * AUX_BYTES
    for k in range(NSLOT):
        S[S_M0SLOT + k] = wave * PW * 1024 + k * CHUNK
    S[S_AUXM0] = LDS_AUX + wave * 1024

    def issue_aux():
        S[S_AG] = S[S_AUXB] + S[S_AUXPOS]
        S[S_AUXPOS] = 0 if S[S_AUXPOS] + AUX_BYTES == S[S_AUXEND] else S[S_AUXPOS] + AUX_BYTES
        st.m0 = S[S_AUXM0]
        dma_aux().emu(st)
        S[S_AUXM0] ^= AUX_BYTES

    def issue_chunk(slot):
        S[S_G] = S[S_W] + S[S_POS]
        S[S_POS] = 0 if S[S_POS] + CHUNK == S[S_END] else S[S_POS] + CHUNK
        st.m0 = S[S_M0SLOT + slot]
        for i in range(PW):
            if i == 4:
                st.m0 += 4096
            dma_piece(i).emu(st)

    issue_aux()
    for k in range(3):
        issue_chunk(k)
    waitcnt_vm(2 * PW).emu(st)
    barrier().emu(st)
    st.A[A_X:A_X + 128] = np.ascontiguousarray(x_tile_regs, dtype=np.float32).view(np.uint32)
    for u in range(15):
        st.run(split_ops(u))
    st.run(pro)
    for b in range(n_block):
        st.run(body)
    waitcnt_lgkm(0).emu(st)
    errs = list(st.errors)
    if check_hazards:
        errs += check_hazards_stream(body + body)
    return st.A[A_X:A_X + 128].view(np.float32).copy(), errs


def check_hazards_stream(stream):
    """static check with a coarse cycle model (other instructions 4 cycles, an MFMA occupies the matrix
    pipe 16 cycles and issues when the pipe is free); the rules are listed in the module docstring"""
    errs = []
    t = 0
    pipe_free = 0
    mf_end = {}       # reg -> end cycle of the MFMA that last wrote it
    valu_wr = {}      # reg -> instruction index of the last VALU write
    mf_rd = {}        # reg -> issue cycle of the last MFMA reading it
    for i, ins in enumerate(stream):
        if i > 0 and stream[i - 1].partial:
            touched = set(ins.rd) | (set(ins.wr) if ins.partial else set())
            if set(stream[i - 1].wr) & touched:
                errs.append('%d: %s touches a register half-written by the instruction right before it '
                            '(dst-sel forwarding)' % (i, ins.text))
        if i > 0 and ins.kind == 'dma' and stream[i - 1].kind == 'salu' and ' m0,' in stream[i - 1].text:
            errs.append('%d: LDS-DMA right behind an M0 write' % i)
        if ins.kind in ('mfma16', 'mfma6'):
            dur = 16
            start = max(t, pipe_free)
            d = set(ins.wr)
            for r in ins.rd:
                if r in d:
                    continue  # C operand = D: accumulate chain
                if r in mf_end and start < mf_end[r] + 24:
                    errs.append('%d: %s reads %s%d too early behind an MFMA' % (i, ins.text, r[0], r[1]))
                if r in valu_wr and i - valu_wr[r] < 3:
                    errs.append('%d: %s reads %s%d written by VALU %d instructions ago' %
                                (i, ins.text, r[0], r[1], i - valu_wr[r]))
                mf_rd[r] = start
            pipe_free = start + dur
            for r in ins.wr:
                mf_end[r] = start + dur
            t = start + 8
        else:
            for r in ins.rd:
                if r in mf_end and t < mf_end[r] + 24:
                    errs.append('%d: %s reads %s%d %d cycles after its MFMA ended' %
                                (i, ins.text, r[0], r[1], t - mf_end[r]))
            for r in ins.wr:
                if ins.kind == 'valu' and r in mf_rd and t < mf_rd[r] + 12:  # (LDS data lands >= 64 cycles later)
                    errs.append('%d: %s overwrites %s%d read by an MFMA %d cycles ago' %
                                (i, ins.text, r[0], r[1], t - mf_rd[r]))
                if r in mf_end and t < mf_end[r]:
                    errs.append('%d: %s overwrites %s%d while an MFMA still writes it' % (i, ins.text, r[0], r[1]))
                if ins.kind == 'valu':
                    valu_wr[r] = i
                mf_end.pop(r, None)
            t += 4 * ins.cost if ins.kind in ('nop',) else 4
    return errs


def model_cycles(body):
    """coarse issue model: cycles of one block"""
    t = pipe = 0
    for ins in body:
        if ins.kind in ('mfma16', 'mfma6'):
            st = max(t, pipe)
            pipe = st + 16
            t = st + 8
        elif ins.kind == 'dma':
            t += 36
        else:
            t += 4 * (ins.cost if ins.kind == 'nop' else 1)
    return max(t, pipe)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--emit', help='write the inline-asm include file')
    ap.add_argument('--dma-burst', action='store_true')
    ap.add_argument('--rd-lead', type=int, default=3)
    ap.add_argument('--cap16', type=int, default=3)
    ap.add_argument('--cap6', type=int, default=3)
    ap.add_argument('--dump', help='write the loop body as plain text')
    ap.add_argument('--skip-terms', default='', help='diagnostics only: comma list of correction terms to drop')
    ap.add_argument('--sim32', action='store_true', help='timing model only: 32x32 MFMA shapes (wrong results)')
    ap.add_argument('--drop', default='', help='diagnostics only: comma list of instruction classes left out of the block loop '
                    '(lgkm, dma, valu, ds, mfma6, mfma16): timing knock-outs, wrong results')
    a = ap.parse_args()
    opts = Opts(dma_burst=a.dma_burst, rd_lead=a.rd_lead, cap16=a.cap16, cap6=a.cap6,
                skip_terms=tuple(int(t) for t in a.skip_terms.split(',') if t), drop=tuple(x for x in a.drop.split(',') if x), sim32=a.sim32)
    if a.emit:
        n = emit_inc(a.emit, opts)
        print('wrote', a.emit, n)
    if a.dump:
        pro, body = steady_block(opts)
        with open(a.dump, 'w') as f:
            for ins in body:
                f.write(ins.text + '\n')
        print('model cycles per block', model_cycles(body))


if __name__ == '__main__':
    sys.exit(main())
