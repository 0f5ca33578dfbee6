#!/usr/bin/env python3
"""Generator of the hand-scheduled gfx950 body loop of the R2L W256 ResMLP (fp16 main pass + two
e4m3 x e5m2 correction terms), plus a lane-accurate CPU emulator of the generated stream.

What is computed (reference: model/nerf_raybased.py:443-465, ResMLP.forward, 43 blocks):
    x <- x + W2 relu(W1 x + b1) + b2            (activations in the act_scale domain)
with b2 folded on the host (x~_i = x_i - sum_{j<i} b2_j, b1'_i = b1_i + W1_i sum_{j<i} b2_j), so a
block is  h = relu(W1 x~ + b1'),  x~ += W2 h : the second layer accumulates IN PLACE into the fp32
residual stream, which is the MFMA C/D operand.

Machine model (one wave64 = 32 rays = 2 column tiles of 16, 4 waves per workgroup, one per SIMD):
  VGPR   0..127  X      fp32 residual stream, X(u,c)+i = feature 16u + 4(lane>>4) + i of ray c*16 + (lane&15)
       128..143  ACC    layer-1 accumulators, 2 buffers x 2 column tiles x 4
       144..151  BIAS   layer-1 bias of a row tile (C operand of its first MFMA), 2 buffers
       152..167  HI     fp16 weight fragments (A operand), 4 buffers
       168..183  A8     e4m3 weight operands (A operand of the K=128 MFMA), 2 buffers x 8
       184..207  TMP    epilogue temporaries, 2 sets x 12
       208..     addresses / constants
  AGPR   0..127  IN     B operands of layer 1: hi fp16 (64) | e5m2 of x (32) | e5m2 of x - hi (32)
       128..255  H      B operands of layer 2, same structure
The weight stream goes global -> LDS ring (4 slots x 32 KiB, LDS-DMA, 3 chunks ahead) -> ds_read_b128.
A chunk = 2 row tiles (32 output features) of one layer: 16 hi fragments + 8 e4m3 operands (32 pieces
of 1 KiB, lane-linear).  One counted vmcnt + one s_barrier per chunk (at its middle).

Every instruction is an `Ins` with its assembly text, the registers it reads / writes and a Python
closure that executes it on the emulator state; `schedule_block` interleaves the fixed MFMA anchor
sequence with the filler instructions (LDS reads, epilogue VALU, LDS-DMA, waits) by a small list
scheduler.  `python body_gen.py --emit r2l_body_asm.inc` writes the inline-asm body; the tests run the
emulator against a float64 reference (tests/test_body_gen_cpu.py).
"""
import argparse
import sys

import numpy as np

# ---------------------------------------------------------------------------------------------
# register map
# ---------------------------------------------------------------------------------------------
NLANE = 64
V_X = 0
V_ACC = 128
V_BIAS = 144
V_HI = 152
V_A8 = 168
V_TMP = 184
V_L0 = 208        # lane*16                (LDS slots 0, 1)
V_L1 = 209        # lane*16 + 65536        (LDS slots 2, 3)
V_AUX = 210       # LDS aux base + (lane>>4)*16 (+1024 on odd blocks)
V_DMAOFF = 211    # wave*8192 + lane*16   (pieces 0..3; 219: + 4096, pieces 4..7)
V_SB = 214        # E8M0 scale 1.0 (activations)
V_AUXOFF = 215    # wave*1024 + lane*16
V_XADDR = 216     # 216,217: 64-bit address for the x tile loads / stores
V_LANE = 218
V_DMAOFF2 = 219
V_SC = 220        # 220,221: E8M0 scales (w - hi | w) of layer 1 of a block; 222,223: of layer 2
N_VGPR_USED = 224

A_IN = 0
A_H = 128

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
S_END = 54     # n_block * 16 * 32768
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

CHUNK = 32768
LDS_AUX = 4 * CHUNK
AUX_BYTES = 4096           # per block: 256 f32 bias | 4 lane quarters x (swl1, sw1, swl2, sw2) | pad
AUX_SCALES = 1024
LDS_BYTES = LDS_AUX + 2 * AUX_BYTES


def layer_exponent(W):
    """e with max|w| in [2^(e-1), 2^e)"""
    m = float(np.abs(W).max())
    return int(np.frexp(m)[1]) if m > 0 else -4


def scale_bytes(e):
    """E8M0 bytes of the two e4m3 operands of a layer with weight exponent e: the operands are stored as
    (w - hi(w)) * 2^(20 - e) and w * 2^(8 - e), both < 2^8 in magnitude"""
    return 127 - (20 - e), 127 - (8 - e)


def X(u, c):
    return V_X + (u * 2 + c) * 4


def ACC(p, c):
    return V_ACC + p * 8 + c * 4


def BIAS(p):
    return V_BIAS + p * 4


def HI(b):
    return V_HI + b * 4


def A8(b):
    return V_A8 + b * 8


def TMP(k):
    return V_TMP + k * 12


def B_hi(base, s, c):
    return base + (s * 2 + c) * 4


def B_a(base, t, c):
    return base + 64 + (t * 2 + c) * 8


def B_r(base, t, c):
    return base + 96 + (t * 2 + c) * 8


# order of the four K=128 MFMAs of a row tile: (term, t); term 0 = (w - hi) x e5m2(a), 1 = w x e5m2(a - hi)
J_ORDER = [(0, 0), (1, 0), (0, 1), (1, 1)]


# ---------------------------------------------------------------------------------------------
# layout maps shared with the host packer (r2l_common.h restated; tests compare both sides)
# ---------------------------------------------------------------------------------------------
def kappa(s, q, j):
    """input feature multiplied by element j of lane quarter q of fp16 k-step s (r2l_kappa)"""
    return 32 * s + 16 * (j >> 2) + 4 * q + (j & 3)


def mix_feat(t, q, e):
    """input feature multiplied by byte e (0..31) of lane quarter q of K=128 step t (r2l_mix_feat)"""
    return 16 * (8 * t + (e >> 2)) + 4 * q + (e & 3)


def piece_hi(upos, s):
    return upos * 8 + s


def piece_a8(upos, j, half):
    return 16 + upos * 8 + j * 2 + half


# ---------------------------------------------------------------------------------------------
# number formats (emulator + python-side packer used by the tests)
# ---------------------------------------------------------------------------------------------
_TABLES = {}


def _fp8_tables():
    if not _TABLES:
        import torch
        b = torch.arange(256, dtype=torch.uint8)
        _TABLES['e4m3'] = b.view(torch.float8_e4m3fn).float().numpy().astype(np.float64)
        _TABLES['e5m2'] = b.view(torch.float8_e5m2).float().numpy().astype(np.float64)
    return _TABLES


def f32_to_e5m2(x):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    return t.to(torch.float8_e5m2).view(torch.uint8).numpy()


def f32_to_e4m3(x):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    return t.to(torch.float8_e4m3fn).view(torch.uint8).numpy()


def pack_body_image(W1s, b1s, W2s, b2s, act_scale=16.0):
    """Python restatement of the host packer (r2l_capi.hip pack_body_v3): returns (stream bytes,
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
            bwl, bw = scale_bytes(ex)
            for qq in range(4):
                aux[b, AUX_SCALES // 4 + 4 * qq + 2 * layer] = 0x01010101 * bwl
                aux[b, AUX_SCALES // 4 + 4 * qq + 2 * layer + 1] = 0x01010101 * bw
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
                        op = np.zeros((64, 32), dtype=np.uint8)
                        for e in range(32):
                            k = mix_feat(t, q, e)
                            w = Wl[rows, k]
                            if term == 0:
                                v = np.ldexp(w - hi[rows, k].astype(np.float32), 20 - ex)
                            else:
                                v = np.ldexp(w, 8 - ex)
                            op[:, e] = f32_to_e4m3(v)
                        for half in range(2):
                            p = base + piece_a8(upos, j, half) * 1024
                            img[p:p + 1024] = op[:, 16 * half:16 * half + 16].reshape(-1)
        Bsum = Bsum + b2s[b].astype(np.float64)
    return img, aux, Bsum


# ---------------------------------------------------------------------------------------------
# instruction objects
# ---------------------------------------------------------------------------------------------
class Ins:
    __slots__ = ('text', 'kind', 'rd', 'wr', 'emu', 'cost', 'tag', 'partial')

    def __init__(self, text, kind, rd=(), wr=(), emu=None, cost=1, tag='', partial=False):
        self.partial = partial  # writes 16 bits of its destination (dst-sel forwarding hazard: see check_hazards_stream)
        self.text = text
        self.kind = kind      # 'mfma16' 'mfma8' 'valu' 'ds' 'dma' 'vmem' 'salu' 'wait' 'barrier' 'nop' 'label' 'branch'
        self.rd = tuple(rd)   # registers read:  ('v', n) / ('a', n)
        self.wr = tuple(wr)
        self.emu = emu
        self.cost = cost      # issue slots (4-cycle units) used by the scheduler's budget
        self.tag = tag


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
    def __init__(self, wave, img, aux, xin, n_block):
        self.V = np.zeros((256, NLANE), dtype=np.uint32)
        self.A = np.zeros((256, NLANE), dtype=np.uint32)
        self.S = {}
        self.lds = np.zeros(LDS_BYTES, dtype=np.uint8)
        self.m0 = 0
        self.wave = wave
        self.img = img                      # uint8 weight stream
        self.aux = aux.view(np.uint8).reshape(-1)   # uint8 view of [n_block, 1024] dwords
        self.xin = xin                      # uint8 view
        self.xout = np.zeros_like(xin)
        self.n_block = n_block
        self.pend_ds = []                   # [(regs, data)] in issue order
        self.pend_regs = set()
        self.pend_dma = []                  # [(list of (lds_addr, bytes))] in issue order
        self.cert = None                    # N of the last vmcnt wait
        self.lds_pending = np.zeros(LDS_BYTES, dtype=bool)
        self.n_ins = 0
        self.errors = []
        self.vm_other = 0                   # plain VMEM ops issued (x loads/stores) since the last vmcnt(0)

    def f32(self, file, n):
        return (self.V if file == 'v' else self.A)[n].view(np.float32)

    def check_rd(self, ins):
        for r in ins.rd:
            if r in self.pend_regs:
                self.errors.append('ins %d (%s) reads %s%d before its ds_read was waited for' %
                                   (self.n_ins, ins.text, r[0], r[1]))

    def run(self, stream):
        for ins in stream:
            if ins.kind in ('label',):
                continue
            self.check_rd(ins)
            if ins.emu is not None:
                ins.emu(self)
            self.n_ins += 1


# ---- builders ----------------------------------------------------------------------------------
def _halves(regs):
    """[n, 64] uint32 -> [64, 2n] float32 of the packed f16 halves (low half first)"""
    h = regs.T.copy().view(np.float16)   # [64, 2n]
    return h.astype(np.float32)


def mfma16(d, a, b_agpr, c, tag=''):
    """v[d:d+3] = A(v[a:a+3]) x B(a[b:b+3]) + v[c:c+3]"""
    text = 'v_mfma_f32_16x16x32_f16 %s, %s, %s, %s' % (vreg(d, 4), vreg(a, 4), areg(b_agpr, 4), vreg(c, 4))

    def emu(st):
        lanes = np.arange(64)
        Ah = _halves(st.V[a:a + 4])         # [64, 8]
        Bh = _halves(st.A[b_agpr:b_agpr + 4])
        Am = np.zeros((16, 32))
        Bm = np.zeros((32, 16))
        for j in range(8):
            Am[lanes & 15, 8 * (lanes >> 4) + j] = Ah[:, j]
            Bm[8 * (lanes >> 4) + j, lanes & 15] = Bh[:, j]
        D = Am @ Bm
        C = st.V[c:c + 4].view(np.float32).astype(np.float64)   # [4, 64]
        out = np.zeros((4, 64), dtype=np.float32)
        for i in range(4):
            out[i] = (C[i] + D[4 * (lanes >> 4) + i, lanes & 15]).astype(np.float32)
        st.V[d:d + 4] = out.view(np.uint32)

    return Ins(text, 'mfma16', rd=vr(a, 4) + ar(b_agpr, 4) + vr(c, 4), wr=vr(d, 4), emu=emu, tag=tag)


def mfma8(d, a, b_agpr, scale_a, tag=''):
    """v[d:d+3] += A(e4m3 v[a:a+7], scale v[scale_a]) x B(e5m2 a[b:b+7])"""
    text = ('v_mfma_scale_f32_16x16x128_f8f6f4 %s, %s, %s, %s, %s, %s op_sel_hi:[0,0,0] blgp:1' %
            (vreg(d, 4), vreg(a, 8), areg(b_agpr, 8), vreg(d, 4), vreg(scale_a), vreg(V_SB)))

    def emu(st):
        T = _fp8_tables()
        lanes = np.arange(64)
        Ab = st.V[a:a + 8].T.copy().view(np.uint8)      # [64, 32]
        Bb = st.A[b_agpr:b_agpr + 8].T.copy().view(np.uint8)
        sa = 2.0 ** (int(st.V[scale_a][0] & 0xff) - 127)
        sb = 2.0 ** (int(st.V[V_SB][0] & 0xff) - 127)
        Am = np.zeros((16, 128))
        Bm = np.zeros((128, 16))
        for e in range(32):
            Am[lanes & 15, 32 * (lanes >> 4) + e] = T['e4m3'][Ab[:, e]] * sa
            Bm[32 * (lanes >> 4) + e, lanes & 15] = T['e5m2'][Bb[:, e]] * sb
        D = Am @ Bm
        C = st.V[d:d + 4].view(np.float32).astype(np.float64)
        out = np.zeros((4, 64), dtype=np.float32)
        for i in range(4):
            out[i] = (C[i] + D[4 * (lanes >> 4) + i, lanes & 15]).astype(np.float32)
        st.V[d:d + 4] = out.view(np.uint32)

    return Ins(text, 'mfma8', rd=vr(a, 8) + ar(b_agpr, 8) + vr(d, 4) + vr(scale_a) + vr(V_SB), wr=vr(d, 4), emu=emu,
               tag=tag)


def ds_read_b128(dst, base_v, off, tag=''):
    assert 0 <= off < 65536 and off % 16 == 0
    text = 'ds_read_b128 %s, %s offset:%d' % (vreg(dst, 4), vreg(base_v), off)

    def emu(st):
        addr = st.V[base_v].astype(np.int64) + off
        data = np.zeros((4, 64), dtype=np.uint32)
        for l in range(64):
            a0 = int(addr[l])
            if st.lds_pending[a0:a0 + 16].any():
                st.errors.append('ins %d (%s): LDS bytes at %d read before their LDS-DMA was certified' %
                                 (st.n_ins, text, a0))
            data[:, l] = st.lds[a0:a0 + 16].view(np.uint32)
        regs = vr(dst, 4)
        st.pend_ds.append((dst, data))
        st.pend_regs.update(regs)

    return Ins(text, 'ds', rd=vr(base_v), wr=vr(dst, 4), emu=emu, tag=tag)


def ds_read_b64(dst, base_v, off, tag=''):
    text = 'ds_read_b64 %s, %s offset:%d' % (vreg(dst, 2), vreg(base_v), off)

    def emu(st):
        addr = st.V[base_v].astype(np.int64) + off
        data = np.zeros((2, 64), dtype=np.uint32)
        for l in range(64):
            a0 = int(addr[l])
            if st.lds_pending[a0:a0 + 8].any():
                st.errors.append('ins %d (%s): LDS bytes at %d read before their LDS-DMA was certified' %
                                 (st.n_ins, text, a0))
            data[:, l] = st.lds[a0:a0 + 8].view(np.uint32)
        st.pend_ds.append((dst, data))
        st.pend_regs.update(vr(dst, 2))

    return Ins(text, 'ds', rd=vr(base_v), wr=vr(dst, 2), emu=emu, tag=tag)


def waitcnt_lgkm(n):
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
        if n == 0:
            st.vm_other = 0
            _land_dma(st, 0)   # this wave's own; other waves' land at the barrier -- conservative: only at barrier
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
        _land_dma(st, st.cert)
    return Ins('s_barrier', 'barrier', emu=emu)


def valu(text, rd, wr, emu, tag=''):
    return Ins(text, 'valu', rd=rd, wr=wr, emu=emu, tag=tag)


def v_max0(dst, src):
    def emu(st):
        st.V[dst] = np.maximum(st.f32('v', src), np.float32(0)).view(np.uint32)
    return valu('v_max_f32 %s, 0, %s' % (vreg(dst), vreg(src)), vr(src), vr(dst), emu)


def v_cvt_pk_f16(dst, a, b):
    def emu(st):
        lo = st.f32('v', a).astype(np.float16).view(np.uint16).astype(np.uint32)
        hi = st.f32('v', b).astype(np.float16).view(np.uint16).astype(np.uint32)
        st.V[dst] = lo | (hi << 16)
    return valu('v_cvt_pk_f16_f32 %s, %s, %s' % (vreg(dst), vreg(a), vreg(b)), vr(a) + vr(b), vr(dst), emu)


def v_resid(dst, hpk, half, t):
    """dst = t - (float)half(hpk): v_fma_mix_f32 dst, hpk.f16[half], -1.0 (SGPR), t"""
    sel = ' op_sel:[1,0,0]' if half else ''
    text = 'v_fma_mix_f32 %s, %s, %s, %s%s op_sel_hi:[1,0,0]' % (vreg(dst), vreg(hpk), sreg(S_NEG1), vreg(t), sel)

    def emu(st):
        h = ((st.V[hpk] >> (16 * half)) & 0xffff).astype(np.uint16).view(np.float16).astype(np.float32)
        st.V[dst] = (st.f32('v', t) - h).astype(np.float32).view(np.uint32)
    return valu(text, vr(hpk) + vr(t), vr(dst), emu)


def v_cvt_pk_bf8(dst, a, b, high):
    sel = ' op_sel:[0,0,1]' if high else ''
    text = 'v_cvt_pk_bf8_f32 %s, %s, %s%s' % (vreg(dst), vreg(a), vreg(b), sel)

    def emu(st):
        b0 = f32_to_e5m2(st.f32('v', a)).astype(np.uint32)
        b1 = f32_to_e5m2(st.f32('v', b)).astype(np.uint32)
        w = b0 | (b1 << 8)
        if high:
            st.V[dst] = (st.V[dst] & 0x0000ffff) | (w << 16)
        else:
            st.V[dst] = (st.V[dst] & 0xffff0000) | w
    ins = valu(text, vr(a) + vr(b), vr(dst), emu)
    ins.partial = True
    return ins


def v_accw(adst, vsrc):
    def emu(st):
        st.A[adst] = st.V[vsrc]
    return valu('v_accvgpr_write_b32 %s, %s' % (areg(adst), vreg(vsrc)), vr(vsrc), ar(adst), emu)


def s_nop(n):
    return Ins('s_nop %d' % n, 'nop', cost=n + 1)


def salu(text, emu=None):
    return Ins(text, 'salu', emu=emu)


def dma_piece(i, tag=''):
    """piece i (0..7) of this wave's 8 KiB of a chunk: global_load_lds_dwordx4 v_off, s[S_G:S_G+1] offset:imm
    with LDS destination M0 + imm + lane*16; pieces 4..7 use the +4096 offset register and M0 + 4096."""
    voffr = V_DMAOFF if i < 4 else V_DMAOFF2
    imm = 1024 * (i & 3)
    text = 'global_load_lds_dwordx4 %s, %s offset:%d' % (vreg(voffr), sreg(S_G, 2), imm)

    def emu(st):
        copies = []
        g = st.S[S_G]
        for w in range(4):
            dw = (w - st.wave) * 8192
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
        self.ins = ins            # Ins or a callable(sched) -> Ins (late-bound waits)
        self.earliest = earliest  # may be issued after anchor #earliest has been emitted
        self.deadline = deadline  # must be issued before anchor #deadline
        self.chain = chain        # fillers of one chain keep their order
        self.seq = 0


def slot_of(layer, m):
    return (layer * 8 + m) % 4


def lds_base_off(slot, piece):
    """(base VGPR, immediate offset) of piece `piece` of ring slot `slot`"""
    off = slot * CHUNK + piece * 1024
    return (V_L0, off) if off < 65536 else (V_L1, off - 65536)


def tile_anchors(T):
    """the 24 MFMAs of tile T of a block as (kind, s_or_j, c); tile T: layer T>>4, row tile T&15"""
    out = []
    for s in range(8):
        out.append(('m16', s, 0))
        out.append(('m16', s, 1))
        if s & 1:
            out.append(('m8', s >> 1, 0))
            out.append(('m8', s >> 1, 1))
    return out


ANCH_PER_TILE = 24


def anchor_index(T, kind, sj, c):
    """global anchor number of an MFMA (T may be <0 or >=32: neighbouring block iterations)"""
    if kind == 'm16':
        k = sj * 2 + c + 2 * (sj // 2)  # sj // 2 pairs of K=128 MFMAs precede k-step sj
    else:
        k = (2 * sj + 1) * 2 + 2 + c + 2 * sj  # after m16 pair of s = 2 sj + 1
    return T * ANCH_PER_TILE + k


def epilogue_ops(T, c, k):
    """VALU epilogue of row tile T (block-local tile index, may be -1 = tile 31 of the previous
    block) for column tile c with temp set k.  Returns [(Ins, consumer)], consumer in {None,'hi','a','r'}"""
    Tm = T % 32
    layer, u = Tm >> 4, Tm & 15
    tb = TMP(k)
    t = [tb + i for i in range(4)]
    h01, h23 = tb + 4, tb + 5
    l = [tb + 6 + i for i in range(4)]
    qa, qr = tb + 10, tb + 11
    ops = []
    if layer == 0:
        src = [ACC(T & 1, c) + i for i in range(4)]
        dst_base = A_H
        for i in range(4):
            ops.append((v_max0(t[i], src[i]), None))
    else:
        t = [X(u, c) + i for i in range(4)]
        dst_base = A_IN
    # gfx950 dst-sel forwarding hazard: the instruction right behind a VALU that writes half a register
    # (v_cvt_pk_bf8_f32) must not read that register -> an independent instruction of the chain sits in between
    ops.append((v_cvt_pk_f16(h01, t[0], t[1]), None))
    ops.append((v_cvt_pk_f16(h23, t[2], t[3]), None))
    ops.append((v_cvt_pk_bf8(qa, t[0], t[1], False), None))
    ops.append((v_cvt_pk_bf8(qa, t[2], t[3], True), None))
    ops.append((v_accw(B_hi(dst_base, u >> 1, c) + 2 * (u & 1), h01), 'hi'))
    ops.append((v_accw(B_hi(dst_base, u >> 1, c) + 2 * (u & 1) + 1, h23), 'hi'))
    ops.append((v_resid(l[0], h01, 0, t[0]), None))
    ops.append((v_resid(l[1], h01, 1, t[1]), None))
    ops.append((v_resid(l[2], h23, 0, t[2]), None))
    ops.append((v_resid(l[3], h23, 1, t[3]), None))
    ops.append((v_cvt_pk_bf8(qr, l[0], l[1], False), None))
    ops.append((v_cvt_pk_bf8(qr, l[2], l[3], True), None))
    ops.append((v_accw(B_a(dst_base, u >> 3, c) + (u & 7), qa), 'a'))
    ops.append((v_accw(B_r(dst_base, u >> 3, c) + (u & 7), qr), 'r'))
    return ops


class Sched:
    """Emits the instruction list of `n_iter` consecutive block iterations; iteration 1 of 3 is the
    steady-state loop body."""

    def __init__(self, opts):
        self.o = opts
        self.out = []             # (iteration, Ins)
        self.ds_issued = 0        # LDS reads issued so far (global count)
        self.ds_done = 0          # all LDS reads with index < ds_done are known complete
        self.ds_index = {}        # key -> index of its LAST ds_read

    # -- LDS read bookkeeping: counted lgkmcnt ------------------------------------------------
    def emit(self, it, ins):
        self.out.append((it, ins))
        if ins.kind == 'ds':
            self.ds_issued += 1

    def need(self, it, key):
        """make sure the LDS reads registered under `key` have landed"""
        if key not in self.ds_index and it == 0:
            return  # issued by the iteration before the schedule starts (iteration 0 is never the extracted one)
        idx = self.ds_index[key]
        if idx < self.ds_done:
            return
        n_after = self.ds_issued - idx - 1
        self.emit(it, waitcnt_lgkm(n_after))
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
        # --- weight operand reads, one group per k-step pair p (k-steps 2p, 2p+1 + the K=128 operand j = p):
        # issue order [bias] a8 lo, a8 hi, hi(2p+1), hi(2p), so that the single counted wait in front of
        # k-step 2p covers the whole group.  hi buffers: pair (p & 1) of 4; a8 buffers: (T*4 + p) & 1.
        for p in range(4):
            g0 = T * 8 + 2 * p
            gj = T * 4 + p
            gp = g0 - opts.rd_lead          # the pair was last used by k-steps g0-4, g0-3; a8 by M8 gj-2 (k-step g0-3)
            earliest = max(A(gp // 8, 'm16', gp % 8, 1), A((gj - 2) // 4, 'm8', (gj - 2) % 4, 1))
            deadline = A(T, 'm16', 2 * p, 0)
            grp = []
            if layer == 0 and p == 0:
                grp.append(ds_read_b128(BIAS(T & 1), V_AUX, 64 * u, tag=('bias', it, T)))
            for half in range(2):
                bv, off = lds_base_off(slot, piece_a8(upos, p, half))
                grp.append(ds_read_b128(A8(gj & 1) + 4 * half, bv, off, tag=('a8', it, T, p, half)))
            for s_ in (2 * p + 1, 2 * p):
                bv, off = lds_base_off(slot, piece_hi(upos, s_))
                grp.append(ds_read_b128(HI(s_ & 3), bv, off, tag=('hi', it, T, s_)))
            for ins in grp:
                F.append(Filler(ins, earliest, deadline, ('rd',)))
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
        for c in range(2):
            ops = epilogue_ops(Tprev, c, c)
            e0 = A(T, 'm16', 0, 1) + 1 + c  # two further MFMAs behind the last writer of its accumulator
            pl, pu = (Tprev % 32) >> 4, (Tprev % 32) & 15
            for ins, cons in ops:
                dl = A(T + 1, 'm16', 0, 0)  # latest: the accumulator buffer is reused by tile T+1
                if cons is not None:
                    # first MFMA that reads the written B register: next layer (same iteration or next)
                    nl_T0 = (Tprev - pu) + 16  # first tile of the consuming layer, block-local
                    if cons == 'hi':
                        first = A(nl_T0, 'm16', pu >> 1, 0)
                    else:
                        term = 0 if cons == 'a' else 1
                        first = A(nl_T0, 'm8', J_ORDER.index((term, pu >> 3)), 0)
                    dl = min(dl, first - 2)
                F.append(Filler(ins, e0, dl, ('epi', c)))
        # --- rendezvous + refill at the middle of each chunk (start of the upos = 1 tile) -------
        if upos == 1:
            a0 = A(T, 'm16', 0, 0)
            ch = ('dma',)
            F.append(Filler(waitcnt_vm(8), a0 - 1, a0 + 1, ch))
            F.append(Filler(barrier(), a0 - 1, a0 + 1, ch))
            seq = []
            cidx = layer * 8 + m          # chunk of the block being consumed
            tgt_slot = (cidx + 3) % 4
            seq.append(salu('s_add_u32 %s, %s, %s' % (sreg(S_G), sreg(S_W), sreg(S_POS)),
                            lambda st: st.S.__setitem__(S_G, st.S[S_W] + st.S[S_POS])))
            seq.append(salu('s_addc_u32 %s, %s, 0' % (sreg(S_G + 1), sreg(S_W + 1))))
            seq.append(salu('s_add_u32 %s, %s, 0x%x' % (sreg(S_POS), sreg(S_POS), CHUNK),
                            lambda st: st.S.__setitem__(S_POS, st.S[S_POS] + CHUNK)))
            seq.append(salu('s_cmp_eq_u32 %s, %s' % (sreg(S_POS), sreg(S_END))))
            seq.append(salu('s_cselect_b32 %s, 0, %s' % (sreg(S_POS), sreg(S_POS)),
                            lambda st: st.S.__setitem__(S_POS, 0 if st.S[S_POS] == st.S[S_END] else st.S[S_POS])))
            if cidx == 0:
                # bias block of the NEXT block -> the other aux slot; older than this chunk's 8 pieces
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
            for i in range(8):
                if i == 4:
                    seq.append(salu('s_add_u32 m0, m0, 0x1000', lambda st: setattr(st, 'm0', st.m0 + 4096)))
                    seq.append(s_nop(0))
                seq.append(dma_piece(i, tag=('dma', it, T, i)))
            end = A(T, 'm8', 3, 1)
            if opts.dma_burst:
                for ins in seq:
                    F.append(Filler(ins, a0 - 1, a0 + 1, ch))
            else:
                # SALU prelude right behind the barrier, then one piece behind each K=128 MFMA
                first_piece = next(i for i, x in enumerate(seq) if x.kind == 'dma' and x.cost == 8)
                for ins in seq[:first_piece]:
                    F.append(Filler(ins, a0 - 1, a0 + 4, ch))
                m8s = [A(T, 'm8', j, c) for j in range(4) for c in range(2)]
                k = 0
                for ins in seq[first_piece:]:
                    F.append(Filler(ins, m8s[min(k, 7)], end + 1, ch))
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
    # fillers that belong before the first anchor (prefetch for tile 0 of iteration 0, epilogue of
    # "tile -1") are scheduled in a virtual pre-region at anchor -1
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

    # pre-region
    pos = -1
    while True:
        r = ready(pos)
        if not r:
            break
        issue(r[0], -1)

    for a in range(total_anchors):
        it = a // (32 * ANCH_PER_TILE)
        T = (a // ANCH_PER_TILE) % 32
        kind, sj, c = tile_anchors(T)[a % ANCH_PER_TILE]
        # forced fillers: deadline reached
        while True:
            r = [f for f in ready(a - 1) if f.deadline <= a]
            if not r:
                break
            issue(r[0], it)
        layer, u = T >> 4, T & 15
        if kind == 'm16':
            g = T * 8 + sj
            sch.need(it, ('hi', it, T, sj))
            in_base = A_IN if layer == 0 else A_H
            if layer == 0:
                d = ACC(T & 1, c)
                csrc = BIAS(T & 1) if sj == 0 else d
                if sj == 0:
                    sch.need(it, ('bias', it, T))
            else:
                d = X(u, c)
                csrc = d
            ins = mfma16(d, HI(sj & 3), B_hi(in_base, sj, c), csrc, tag=('m16', it, T, sj, c))
            cap = opts.cap16
        else:
            gj = T * 4 + sj
            sch.need(it, ('a8', it, T, sj, 1))
            term, t = J_ORDER[sj]
            in_base = A_IN if layer == 0 else A_H
            d = ACC(T & 1, c) if layer == 0 else X(u, c)
            bop = B_a(in_base, t, c) if term == 0 else B_r(in_base, t, c)
            if T & 15 == 0 and sj == 0 and c == 0:
                sch.need(it, ('scale', it, layer))
            ins = mfma8(d, A8(gj & 1), bop, V_SC + 2 * layer + term, tag=('m8', it, T, sj, c))
            cap = opts.cap8
        if not (kind == 'm8' and J_ORDER[sj][0] in opts.skip_terms):
            sch.emit(it, ins)
        budget = cap
        while budget > 0:
            r = ready(a)
            if not r:
                break
            f = r[0]
            issue(f, it)
            budget -= f.ins.cost
    # leftovers (belong to iterations beyond the schedule): dropped
    return sch.out


class Opts:
    def __init__(self, **kw):
        self.rd_lead = 3
        self.cap16 = 2
        self.cap8 = 6
        self.dma_burst = False
        self.skip_terms = ()      # diagnostics: drop the K=128 MFMAs of these correction terms (wrong results)
        self.__dict__.update(kw)


def steady_block(opts):
    """(prologue reads, loop body) -- the body is iteration 1 of a 3-iteration schedule; the prologue is
    the set of LDS reads of iteration 1 that the schedule placed inside iteration 0 (re-issued before
    the loop is entered), in their issue order."""
    out = schedule(opts, 3)
    body = [ins for it, ins in out if it == 1]
    pro = []
    for it, ins in out:
        if it == 0 and ins.kind == 'ds' and ins.tag[1] == 1:
            pro.append(ins)
    return pro, body


# ---------------------------------------------------------------------------------------------
# whole-kernel text
# ---------------------------------------------------------------------------------------------
def split_ops(u):
    """standalone split of X row tile u -> IN set (the layer-2 epilogue without MFMAs)"""
    ops = []
    for c in range(2):
        ops += [ins for ins, _ in epilogue_ops(16 + u, c, c)]
    return ops


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
    a('v_lshrrev_b32 %s, 4, %s' % (vreg(V_AUX), vreg(V_LANE)))
    a('v_lshlrev_b32 %s, 4, %s' % (vreg(V_AUX), vreg(V_AUX)))
    a('v_add_u32 %s, 0x%x, %s' % (vreg(V_AUX), LDS_AUX, vreg(V_AUX)))
    a('s_lshl_b32 %s, %s, 13' % (sreg(S_T0), sreg(S_WAVE)))          # wave * 8192
    a('v_add_u32 %s, %s, %s' % (vreg(V_DMAOFF), sreg(S_T0), vreg(V_L0)))
    a('v_add_u32 %s, 0x1000, %s' % (vreg(V_DMAOFF2), vreg(V_DMAOFF)))
    a('s_lshl_b32 %s, %s, 10' % (sreg(S_T0 + 1), sreg(S_WAVE)))       # wave * 1024
    a('v_add_u32 %s, %s, %s' % (vreg(V_AUXOFF), sreg(S_T0 + 1), vreg(V_L0)))
    for k in range(4):
        a('s_add_u32 %s, %s, 0x%x' % (sreg(S_M0SLOT + k), sreg(S_T0), k * CHUNK))
    a('s_add_u32 %s, %s, 0x%x' % (sreg(S_AUXM0), sreg(S_T0 + 1), LDS_AUX))
    a('v_mov_b32 %s, 0x7f7f7f7f' % vreg(V_SB))
    a('s_lshl_b32 %s, %s, 19' % (sreg(S_END), sreg(S_NBLOCK)))        # n_block * 16 * 32768
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
        for i in range(8):
            if i == 4:
                r += ['s_add_u32 m0, m0, 0x1000', 's_nop 0']
            r.append(dma_piece(i).text)
        return r

    L += issue_aux
    for k in range(3):
        L += issue_chunk(k)
    a('s_waitcnt vmcnt(16)')
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
        a('global_load_dwordx4 %s, %s, %s offset:%d' % (vreg(X(0, 0) + 4 * i, 4), vreg(V_L0), sreg(S_T0 + 4, 2),
                                                        (i % 4) * 1024))
        if i % 4 == 3:
            a('s_add_u32 %s, %s, 0x1000' % (sreg(S_T0 + 4), sreg(S_T0 + 4)))
            a('s_addc_u32 %s, %s, 0' % (sreg(S_T0 + 5), sreg(S_T0 + 5)))
    a('s_waitcnt vmcnt(0)')
    # initial split of row tiles 0..14 (tile 15's runs at the head of the loop body)
    for u in range(15):
        for ins in split_ops(u):
            a(ins.text)
    # LDS reads the loop head expects in flight
    for ins in pro:
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
        a('global_store_dwordx4 %s, %s, %s offset:%d' % (vreg(V_L0), vreg(X(0, 0) + 4 * i, 4), sreg(S_T0 + 4, 2),
                                                         (i % 4) * 1024))
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
    st = State(wave, img, aux, np.zeros(4, dtype=np.uint8), n_block)
    lanes = np.arange(64, dtype=np.uint32)
    st.V[V_LANE] = lanes
    st.V[V_L0] = lanes * 16
    st.V[V_L1] = lanes * 16 + 65536
    st.V[V_AUX] = LDS_AUX + (lanes >> 4) * 16
    st.V[V_DMAOFF] = wave * 8192 + lanes * 16
    st.V[V_DMAOFF2] = wave * 8192 + lanes * 16 + 4096
    st.V[V_AUXOFF] = wave * 1024 + lanes * 16
    st.V[V_SB] = 0x7f7f7f7f
    S = st.S
    S[S_W] = 0
    S[S_AUXB] = 0
    S[S_POS] = 0
    S[S_AUXPOS] = 0
    S[S_END] = n_block * 16 * CHUNK
    S[S_AUXEND] = n_block * AUX_BYTES
    for k in range(4):
        S[S_M0SLOT + k] = wave * 8192 + k * CHUNK
    S[S_AUXM0] = LDS_AUX + wave * 1024

    def run_salu_issue_aux():
        S[S_AG] = S[S_AUXB] + S[S_AUXPOS]
        S[S_AUXPOS] = 0 if S[S_AUXPOS] + AUX_BYTES == S[S_AUXEND] else S[S_AUXPOS] + AUX_BYTES
        st.m0 = S[S_AUXM0]
        dma_aux().emu(st)
        S[S_AUXM0] ^= AUX_BYTES

    def run_issue_chunk(slot):
        S[S_G] = S[S_W] + S[S_POS]
        S[S_POS] = 0 if S[S_POS] + CHUNK == S[S_END] else S[S_POS] + CHUNK
        st.m0 = S[S_M0SLOT + slot]
        for i in range(8):
            if i == 4:
                st.m0 += 4096
            dma_piece(i).emu(st)

    run_salu_issue_aux()
    for k in range(3):
        run_issue_chunk(k)
    waitcnt_vm(16).emu(st)
    barrier().emu(st)
    st.V[V_X:V_X + 128] = np.ascontiguousarray(x_tile_regs, dtype=np.float32).view(np.uint32)
    for u in range(15):
        st.run(split_ops(u))
    st.run(pro)
    for b in range(n_block):
        st.run(body)
    waitcnt_lgkm(0).emu(st)
    errs = list(st.errors)
    if check_hazards:
        errs += check_hazards_stream(body + body)
    return st.V[V_X:V_X + 128].view(np.float32).copy(), errs


def check_hazards_stream(stream):
    """static check with a coarse cycle model (other instructions 4 cycles, an MFMA occupies the matrix
    pipe 16 / 32 cycles and issues when the pipe is free):
      * a non-MFMA read (or an MFMA A/B read) of an MFMA result happens >= 24 cycles after that MFMA ended
      * a VALU write is read by an MFMA as A/B no earlier than 3 instructions later
      * an LDS / VALU write never lands on a register an MFMA issued < 12 cycles ago reads as C
      * the instruction right behind a half-register write (v_cvt_pk_bf8_f32) does not read that register: on gfx950
        the forwarding path hands over the stale half (hipcc pads this "dst-sel forwarding" hazard itself)
    """
    errs = []
    t = 0
    pipe_free = 0
    mf_end = {}       # reg -> end cycle of the MFMA that last wrote it
    valu_wr = {}      # reg -> instruction index of the last VALU write
    mf_rd = {}        # reg -> issue cycle of the last MFMA reading it
    for i, ins in enumerate(stream):
        if i > 0 and stream[i - 1].partial and set(stream[i - 1].wr) & set(ins.rd):
            errs.append('%d: %s reads a register half-written by the instruction right before it (dst-sel forwarding)' %
                        (i, ins.text))
        if ins.kind in ('mfma16', 'mfma8'):
            dur = 16 if ins.kind == 'mfma16' else 32
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--emit', help='write the inline-asm include file')
    ap.add_argument('--dma-burst', action='store_true')
    ap.add_argument('--rd-lead', type=int, default=3)
    ap.add_argument('--cap16', type=int, default=2)
    ap.add_argument('--cap8', type=int, default=6)
    ap.add_argument('--dump', help='write the loop body as plain text')
    ap.add_argument('--skip-terms', default='', help='diagnostics only: comma list of correction terms to drop')
    a = ap.parse_args()
    opts = Opts(dma_burst=a.dma_burst, rd_lead=a.rd_lead, cap16=a.cap16, cap8=a.cap8,
                skip_terms=tuple(int(t) for t in a.skip_terms.split(',') if t))
    if a.emit:
        n = emit_inc(a.emit, opts)
        print('wrote', a.emit, n)
    if a.dump:
        pro, body = steady_block(opts)
        with open(a.dump, 'w') as f:
            for ins in body:
                f.write(ins.text + '\n')


if __name__ == '__main__':
    sys.exit(main())
