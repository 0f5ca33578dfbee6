#!/usr/bin/env python3
"""Instruction objects, number formats, the lane-accurate emulator state and the static hazard check shared by the gfx950
stream generators (body_gen.py: R2L ResMLP body, 32x32 MFMA shapes; nerf_gen.py: NeRF teacher layer chain, 16x16 shapes).

Every instruction is an `Ins` with its assembly text, the registers it reads / writes and a Python closure that executes
it on the emulator `State` (64 lanes x 256 VGPRs + 256 AGPRs, LDS bytes, in-order LDS-read and LDS-DMA queues).

Hazards a hand-written stream must respect by construction (hipcc pads nothing inside an asm statement);
`check_hazards_stream` enforces them statically with a coarse cycle model:
  * MFMA result -> any non-accumulate reader: the reader comes well after the MFMA has left the pipe;
  * VALU write -> MFMA operand read: >= 2 instructions in between;
  * a VALU that writes HALF a register (v_fma_mixlo/hi_f16) must not be followed directly by a reader of that register
    ("dst-sel forwarding" hazard: the reader sees the stale half);
  * s_mov m0 -> LDS-DMA: one instruction in between.
"""
import numpy as np

NLANE = 64
BF6_TOP = 4                # 28 = 1.75 * 2^4
# activations (act_scale domain, < 2^7) are converted as a / 2^3, their fp16 residuals (< 2^-5) as r * 2^9
ACT_EXP = 3
RES_EXP = -9


def layer_exponent(W):
    """e with max|w| in [2^(e-1), 2^e)"""
    m = float(np.abs(W).max())
    return int(np.frexp(m)[1]) if m > 0 else -4


def weight_exps(e, fmt='bf6'):
    """power-of-two exponents of the two low-precision weight operands of a layer with weight exponent e:
    bf6: stored (w - hi(w)) / 2^(e-16) and w / 2^(e-4), both < 2^5 in magnitude (bf6 holds 28);
    e4m3: (w - hi(w)) / 2^(e-20) and w / 2^(e-8), both < 2^9 (e4m3 holds 448)"""
    return (e - 16, e - 4) if fmt == 'bf6' else (e - 20, e - 8)


# ---------------------------------------------------------------------------------------------
# layout maps shared with the host packer (r2l_common.h restated; tests compare both sides)
# ---------------------------------------------------------------------------------------------
def kappa16(s, q, j):
    """input feature multiplied by element j of lane quarter q of fp16 k-step s (r2l_kappa)"""
    return 32 * s + 16 * (j >> 2) + 4 * q + (j & 3)


def mix16(t, q, e):
    """input feature multiplied by element e (0..31) of lane quarter q of K=128 step t: the conversion takes the
    16 fp16 pair registers of row tiles 8t .. 8t+7 in order, element e = 4 * (row tile & 7) + accumulator register"""
    return 16 * (8 * t + (e >> 2)) + 4 * q + (e & 3)


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


def _e4m3_table():
    v = np.zeros(128)
    for b in range(128):
        e, m = (b >> 3) & 15, b & 7
        v[b] = (m / 8.0) * 2.0 ** -6 if e == 0 else (1 + m / 8.0) * 2.0 ** (e - 7)
    v[127] = np.inf          # 0x7f is NaN; as a search bound it never wins
    return v


E4M3_POS = _e4m3_table()         # ascending magnitudes of codes 0 .. 126 (448), [127] = inf
E4M3 = np.concatenate([np.where(np.isinf(E4M3_POS), np.nan, E4M3_POS), -np.where(np.isinf(E4M3_POS), np.nan, E4M3_POS)])
E4M3_TOP = 8                     # 448 = 1.75 * 2^8


def f_to_e4m3(x):
    """nearest OCP e4m3 code (ties to even mantissa), saturating at 448; x float array"""
    x = np.asarray(x, dtype=np.float64)
    a = np.abs(x)
    pos = E4M3_POS[:127]
    idx = np.searchsorted(pos, a).clip(1, 126)
    lo, hi = pos[idx - 1], pos[idx]
    up = (a - lo > hi - a) | ((a - lo == hi - a) & (((idx - 1) & 1) == 1))
    code = np.where(up, idx, idx - 1)
    code = np.where(a >= pos[126], 126, code)
    return (code | np.where(np.signbit(x), 128, 0)).astype(np.uint8)


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
    def __init__(self, wave, img, aux, n_block, lds_bytes):
        self.V = np.zeros((256, NLANE), dtype=np.uint32)
        self.A = np.zeros((256, NLANE), dtype=np.uint32)
        self.S = {}
        self.lds = np.zeros(lds_bytes, dtype=np.uint8)
        self.m0 = 0
        self.wave = wave
        self.img = img                              # uint8 weight stream
        self.aux = aux.view(np.uint8).reshape(-1)   # uint8 view of [n_block, 1024] dwords
        self.n_block = n_block
        self.pend_ds = []                   # [(first reg, data)] in issue order
        self.pend_regs = set()
        self.pend_dma = []                  # [(list of (lds_addr, bytes))] in issue order
        self.pend_vload = []                # register loads in flight: [(file, first reg, data, source)] in issue order
        self.pend_stage = []                # staged ds_writes not yet certified by a barrier: [(lgkm token, copies)]
        self.stage_src = {}                 # first staging register -> per-lane global byte address it was loaded from
        self.cert = None                    # N of the last vmcnt wait
        self.lds_pending = np.zeros(lds_bytes, dtype=bool)
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


def _ds_read(width, dst, base_v, off, tag, dfile='v'):
    n = width // 4
    assert 0 <= off < 65536 and off % (16 if width == 12 else width) == 0
    text = 'ds_read_b%d %s, %s offset:%d' % (width * 8, (vreg if dfile == 'v' else areg)(dst, n), vreg(base_v), off)

    def emu(st):
        addr = st.V[base_v].astype(np.int64) + off
        data = np.zeros((n, 64), dtype=np.uint32)
        for l in range(64):
            a0 = int(addr[l])
            if st.lds_pending[a0:a0 + width].any():
                st.errors.append('ins %d (%s): LDS bytes at %d read before their LDS-DMA was certified' %
                                 (st.n_ins, text, a0))
            data[:, l] = st.lds[a0:a0 + width].view(np.uint32)
        st.pend_ds.append((dfile, dst, data))
        st.pend_regs.update(regs)

    regs = vr(dst, n) if dfile == 'v' else ar(dst, n)
    return Ins(text, 'ds', rd=vr(base_v), wr=regs, emu=emu, tag=tag)


def ds_read_b128(dst, base_v, off, tag='', dfile='v'):
    return _ds_read(16, dst, base_v, off, tag, dfile)


def ds_read_b64(dst, base_v, off, tag='', dfile='v'):
    return _ds_read(8, dst, base_v, off, tag, dfile)


def ds_read_b96(dst, base_v, off, tag=''):
    ins = _ds_read(12, dst, base_v, off, tag)
    return ins


def waitcnt_lgkm(n):
    assert 0 <= n <= 15

    def emu(st):
        while len(st.pend_ds) > n:
            dfile, dst, data = st.pend_ds.pop(0)
            st.regs(dfile)[dst:dst + len(data)] = data
            for r in (vr if dfile == 'v' else ar)(dst, len(data)):
                st.pend_regs.discard(r)
    return Ins('s_waitcnt lgkmcnt(%d)' % n, 'wait', emu=emu)


def waitcnt_vm(n):
    def emu(st):
        st.cert = n
        if st.pend_vload and n != 0:
            st.errors.append('ins %d: counted vmcnt(%d) with register loads in flight (not modelled)' % (st.n_ins, n))
        while st.pend_vload and n == 0:
            file, dst, data = st.pend_vload.pop(0)
            st.regs(file)[dst:dst + len(data)] = data
            for r in (vr if file == 'v' else ar)(dst, len(data)):
                st.pend_regs.discard(r)
    return Ins('s_waitcnt vmcnt(%d)' % n, 'wait', emu=emu)


def global_load_x4_a(adst, voff, sbase, imm, tag=''):
    """a[adst:adst+3] <- 16 bytes per lane of the weight stream at s[sbase:sbase+1] + v[voff] + imm (vmcnt)"""
    assert -4096 <= imm < 4096
    text = 'global_load_dwordx4 %s, %s, %s offset:%d' % (areg(adst, 4), vreg(voff), sreg(sbase, 2), imm)

    def emu(st):
        src = st.S[sbase] + st.V[voff].astype(np.int64) + imm
        data = np.zeros((4, 64), dtype=np.uint32)
        for l in range(64):
            assert 0 <= src[l] and src[l] + 16 <= len(st.img), src[l]
            data[:, l] = st.img[int(src[l]):int(src[l]) + 16].view(np.uint32)
        st.pend_vload.append(('a', adst, data))
        st.pend_regs.update(ar(adst, 4))
        st.stage_src[adst] = src.copy()
    return Ins(text, 'vload', rd=vr(voff), wr=ar(adst, 4), emu=emu, tag=tag)


def ds_write_b128_stage(addr_v, asrc, off, wave_bytes, tag=''):
    """one staged piece of a ring slot: ds_write_b128 v[addr_v] + off <- a[asrc:asrc+3].  The three other waves of the
    workgroup do the same with their shares (wave_bytes apart, in the stream and in the slot alike); the bytes count as
    written for everybody behind the next barrier, which this wave may only enter with the write complete (lgkmcnt)."""
    assert 0 <= off < 65536 and off % 16 == 0
    text = 'ds_write_b128 %s, %s offset:%d' % (vreg(addr_v), areg(asrc, 4), off)

    def emu(st):
        addr = st.V[addr_v].astype(np.int64) + off
        src = st.stage_src[asrc]
        copies = []
        for w in range(4):
            dw = (w - st.wave) * wave_bytes
            for l in range(64):
                dst = int(addr[l]) + dw
                if w == st.wave:
                    data = st.A[asrc:asrc + 4, l].copy().view(np.uint8)
                else:
                    data = st.img[int(src[l]) + dw:int(src[l]) + dw + 16].copy()
                copies.append((dst, data))
                st.lds_pending[dst:dst + 16] = True
        token = ('v', 0, np.zeros((0, 64), dtype=np.uint32))      # counts in lgkmcnt like a read
        st.pend_ds.append(token)
        st.pend_stage.append((token, copies))
    return Ins(text, 'ds', rd=vr(addr_v) + ar(asrc, 4), emu=emu, tag=tag)


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
        for token, copies in st.pend_stage:
            if any(token is e for e in st.pend_ds):
                st.errors.append('ins %d: s_barrier with a staged ds_write still in flight (lgkmcnt)' % st.n_ins)
            for addr, data in copies:
                st.lds[addr:addr + len(data)] = data
                st.lds_pending[addr:addr + len(data)] = False
        st.pend_stage = []
    return Ins('s_barrier', 'barrier', emu=emu)


def valu(text, rd, wr, emu, tag='', partial=False):
    return Ins(text, 'valu', rd=rd, wr=wr, emu=emu, tag=tag, partial=partial)


def v_max0(dst, src):
    def emu(st):
        st.V[dst] = np.maximum(st.f32('v', src), np.float32(0)).view(np.uint32)
    return valu('v_max_f32 %s, 0, %s' % (vreg(dst), vreg(src)), vr(src), vr(dst), emu)


def v_mul_lit(dst, src, lit):
    """dst = lit * src, lit an f32 literal (a power of two here: exact)"""
    bits = int(np.float32(lit).view(np.uint32))

    def emu(st):
        st.V[dst] = (st.f32('v', src) * np.float32(lit)).astype(np.float32).view(np.uint32)
    return valu('v_mul_f32 %s, 0x%08x, %s' % (vreg(dst), bits, vreg(src)), vr(src), vr(dst), emu)


def v_fmac_lit(dst, lit, src):
    """dst += lit * src (one rounding), lit an f32 literal"""
    bits = int(np.float32(lit).view(np.uint32))

    def emu(st):
        st.V[dst] = (st.f32('v', dst).astype(np.float64) + np.float64(np.float32(lit)) * st.f32('v', src).astype(np.float64)
                     ).astype(np.float32).view(np.uint32)
    return valu('v_fmac_f32 %s, 0x%08x, %s' % (vreg(dst), bits, vreg(src)), vr(dst) + vr(src), vr(dst), emu)


def v_max3_abs(dst, a, b):
    """dst = max(dst, |a|, |b|) in f32 (dst >= 0): the running maximum of the range guard"""
    def emu(st):
        m = np.maximum(np.abs(st.f32('v', a)), np.abs(st.f32('v', b)))
        st.V[dst] = np.maximum(st.f32('v', dst), m).astype(np.float32).view(np.uint32)
    return valu('v_max3_f32 %s, %s, |%s|, |%s|' % (vreg(dst), vreg(dst), vreg(a), vreg(b)), vr(dst) + vr(a) + vr(b), vr(dst), emu)


def ds_max_u32(addr_v, data_v, off, tag=''):
    """LDS[v[addr_v] + off] = max(LDS[...], v[data_v]) as unsigned (no return value; counts in lgkmcnt like a read)"""
    text = 'ds_max_u32 %s, %s offset:%d' % (vreg(addr_v), vreg(data_v), off)

    def emu(st):
        addr = st.V[addr_v].astype(np.int64) + off
        for l in range(64):
            a0 = int(addr[l])
            cur = st.lds[a0:a0 + 4].view(np.uint32)[0]
            st.lds[a0:a0 + 4] = np.array([max(cur, st.V[data_v][l])], dtype=np.uint32).view(np.uint8)
        st.pend_ds.append(('v', 0, np.zeros((0, 64), dtype=np.uint32)))
    return Ins(text, 'ds', rd=vr(addr_v) + vr(data_v), emu=emu, tag=tag)


def _ds_write(width, addr_v, data_v, off, tag):
    n = width // 4
    assert 0 <= off < 65536 and off % width == 0
    text = 'ds_write_b%d %s, %s offset:%d' % (width * 8, vreg(addr_v), vreg(data_v, n), off)

    def emu(st):
        addr = st.V[addr_v].astype(np.int64) + off
        for l in range(64):
            a0 = int(addr[l])
            st.lds[a0:a0 + width] = st.V[data_v:data_v + n, l].copy().view(np.uint8)
        st.pend_ds.append(('v', 0, np.zeros((0, 64), dtype=np.uint32)))      # counts in lgkmcnt like a read
    return Ins(text, 'ds', rd=vr(addr_v) + vr(data_v, n), emu=emu, tag=tag)


def ds_write_b128(addr_v, data_v, off, tag=''):
    return _ds_write(16, addr_v, data_v, off, tag)


def ds_write_b64(addr_v, data_v, off, tag=''):
    return _ds_write(8, addr_v, data_v, off, tag)


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


def v_resid16(dst, dst_high, hpk, half, t, s_neg1):
    """half `dst_high` of dst = fp16(t - (float)half(hpk)):  v_fma_mixlo/hi_f16 dst, hpk.f16[half], -1.0 (SGPR s_neg1), t"""
    op = 'v_fma_mixhi_f16' if dst_high else 'v_fma_mixlo_f16'
    sel = ' op_sel:[1,0,0]' if half else ''
    text = '%s %s, %s, %s, %s%s op_sel_hi:[1,0,0]' % (op, vreg(dst), vreg(hpk), sreg(s_neg1), vreg(t), sel)

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


def _f32(st, x):
    """operand value: ('v', n) register, ('s', n) scalar register (float bits in st.S as python float), or a float literal"""
    if isinstance(x, tuple):
        if x[0] == 'v':
            return st.V[x[1]].view(np.float32)
        return np.full(64, np.float32(st.S[x[1]]), dtype=np.float32)
    return np.full(64, np.float32(x), dtype=np.float32)


def _opnd(x):
    if isinstance(x, tuple):
        if x[0] == 'v':
            return vreg(x[1])
        return '%%[%s]' % x[1] if isinstance(x[1], str) else sreg(x[1])   # a named scalar = an asm operand of the block
    if x in (0.0, 0.5, 1.0, 2.0, 4.0, -0.5, -1.0, -2.0, -4.0):
        return repr(float(x))
    return '0x%08x' % f32_bits(x)


def _rd(*xs):
    return [x for x in xs if isinstance(x, tuple) and x[0] == 'v']


def v_f32_op(op, dst, a, b, tag=''):
    """dst = a (op) b in f32, one rounding; op in mul / add / sub; a may be a literal / SGPR / VGPR, b a VGPR"""
    fn = {'mul': np.multiply, 'add': np.add, 'sub': np.subtract}[op]

    def emu(st):
        st.V[dst] = fn(_f32(st, a), _f32(st, b)).astype(np.float32).view(np.uint32)
    return valu('v_%s_f32 %s, %s, %s' % (op, vreg(dst), _opnd(a), _opnd(b)), _rd(a, b), vr(dst), emu, tag=tag)


def v_fma_f32(dst, a, b, c, neg_c=False):
    """dst = fma(a, b, +-c): VGPR operands (one rounding)"""
    def emu(st):
        x = _f32(st, a).astype(np.float64) * _f32(st, b).astype(np.float64)    # exact in float64
        st.V[dst] = (x + (-1.0 if neg_c else 1.0) * _f32(st, c).astype(np.float64)).astype(np.float32).view(np.uint32)
    text = 'v_fma_f32 %s, %s, %s, %s%s' % (vreg(dst), _opnd(a), _opnd(b), '-' if neg_c else '', _opnd(c))
    return valu(text, _rd(a, b, c), vr(dst), emu)


def v_ldexp_f32(dst, a, n):
    def emu(st):
        st.V[dst] = np.ldexp(_f32(st, a), n).astype(np.float32).view(np.uint32)
    return valu('v_ldexp_f32 %s, %s, %d' % (vreg(dst), _opnd(a), n), _rd(a), vr(dst), emu)


def v_rndne_f32(dst, a):
    def emu(st):
        st.V[dst] = np.rint(_f32(st, a)).astype(np.float32).view(np.uint32)
    return valu('v_rndne_f32 %s, %s' % (vreg(dst), _opnd(a)), _rd(a), vr(dst), emu)


def v_sin_f32(dst, a):
    """dst = sin(2 pi a): a transcendental (quarter rate); its consumer must not be the next instruction (kind 'trans')"""
    def emu(st):
        st.V[dst] = np.sin(2.0 * np.pi * _f32(st, a).astype(np.float64)).astype(np.float32).view(np.uint32)
    ins = valu('v_sin_f32 %s, %s' % (vreg(dst), _opnd(a)), _rd(a), vr(dst), emu)
    ins.kind = 'trans'
    return ins


def v_mov_b32(dst, a):
    def emu(st):
        st.V[dst] = _f32(st, a).view(np.uint32)
    return valu('v_mov_b32 %s, %s' % (vreg(dst), _opnd(a)), _rd(a), vr(dst), emu)


def v_lshl_or(dst, a, sh, b):
    """dst = (a << sh) | b"""
    def emu(st):
        st.V[dst] = ((st.V[a].astype(np.uint64) << np.uint64(sh)).astype(np.uint32)) | st.V[b]
    return valu('v_lshl_or_b32 %s, %s, %d, %s' % (vreg(dst), vreg(a), sh, vreg(b)), vr(a) + vr(b), vr(dst), emu)


def v_sub_imm(dst, src, imm):
    """dst = src - imm"""
    def emu(st):
        st.V[dst] = (st.V[src].astype(np.int64) - imm).astype(np.uint32)
    return valu('v_subrev_u32 %s, 0x%x, %s' % (vreg(dst), imm, vreg(src)), vr(src), vr(dst), emu)


def v_lshl_imm(dst, sh, src):
    def emu(st):
        st.V[dst] = (st.V[src].astype(np.uint64) << np.uint64(sh)).astype(np.uint32)
    return valu('v_lshlrev_b32 %s, %d, %s' % (vreg(dst), sh, vreg(src)), vr(src), vr(dst), emu)


def s_nop(n):
    return Ins('s_nop %d' % n, 'nop', cost=n + 1)


def salu(text, emu=None):
    return Ins(text, 'salu', emu=emu)




# ---- 32x32 shapes (layouts: tools/mfma32_probe.hip) -------------------------------------------------
# A lane l = row l%32, k-slot (l/32, j); B lane l = column l%32, k-slot (l/32, j); D lane l = column l%32, register r =
# row 8*(r/4) + 4*(l/32) + r%4
def _d32_rows():
    lanes = np.arange(64)
    return np.stack([8 * (r // 4) + 4 * (lanes >> 5) + r % 4 for r in range(16)])   # [16, 64]


_D32_ROWS = _d32_rows()


def mfma32_16(dfile, d, a, b, cfile, c, tag='', bfile='v', afile='v'):
    """D[dfile d:d+15] = A([afile] a:a+3, 32 x 16) x B([bfile] b:b+3, 16 x 32) + C[cfile c:c+15]: v_mfma_f32_32x32x16_f16"""
    rf = {'v': vreg, 'a': areg, '0': lambda c_, n_: '0'}      # cfile '0': the inline constant 0 (a fresh accumulator)
    text = 'v_mfma_f32_32x32x16_f16 %s, %s, %s, %s' % (rf[dfile](d, 16), rf[afile](a, 4), rf[bfile](b, 4), rf[cfile](c, 16))

    def emu(st):
        lanes = np.arange(64)
        Ah = _halves(st.regs(afile)[a:a + 4])         # [64, 8]
        Bh = _halves(st.regs(bfile)[b:b + 4])
        Am = np.zeros((32, 16))
        Bm = np.zeros((16, 32))
        for j in range(8):
            Am[lanes & 31, 8 * (lanes >> 5) + j] = Ah[:, j]
            Bm[8 * (lanes >> 5) + j, lanes & 31] = Bh[:, j]
        D = Am @ Bm
        C = np.zeros((16, 64)) if cfile == '0' else st.regs(cfile)[c:c + 16].view(np.float32).astype(np.float64)   # [16, 64]
        out = np.zeros((16, 64), dtype=np.float32)
        for r in range(16):
            out[r] = (C[r] + D[_D32_ROWS[r], lanes & 31]).astype(np.float32)
        st.regs(dfile)[d:d + 16] = out.view(np.uint32)

    rc = [] if cfile == '0' else vr(c, 16) if cfile == 'v' else ar(c, 16)
    wd = vr(d, 16) if dfile == 'v' else ar(d, 16)
    rb = vr(b, 4) if bfile == 'v' else ar(b, 4)
    ra = vr(a, 4) if afile == 'v' else ar(a, 4)
    return Ins(text, 'mfma16', rd=ra + rb + rc, wr=wd, emu=emu, tag=tag, cost=1)


def mfma32_6(dfile, d, a, b_agpr, scale_a, scale_b, tag='', bfile='a', afile='v'):
    """D += A(bf6 [afile] a:a+5, 32 x 64, E8M0 v[scale_a]) x B(bf6 [bfile] b:b+5, 64 x 32, E8M0 v[scale_b]):
    v_mfma_scale_f32_32x32x64_f8f6f4"""
    rf = {'v': vreg, 'a': areg}
    text = ('v_mfma_scale_f32_32x32x64_f8f6f4 %s, %s, %s, %s, %s, %s op_sel_hi:[0,0,0] cbsz:3 blgp:3' %
            (rf[dfile](d, 16), rf[afile](a, 6), rf[bfile](b_agpr, 6), rf[dfile](d, 16), vreg(scale_a), vreg(scale_b)))

    def emu(st):
        lanes = np.arange(64)
        Ac = unpack6(st.regs(afile)[a:a + 6].T.copy())          # [64, 32]
        Bc = unpack6(st.regs(bfile)[b_agpr:b_agpr + 6].T.copy())
        sa = 2.0 ** (int(st.V[scale_a][0] & 0xff) - 127)
        sb = 2.0 ** (int(st.V[scale_b][0] & 0xff) - 127)
        Am = np.zeros((32, 64))
        Bm = np.zeros((64, 32))
        for e in range(32):
            Am[lanes & 31, 32 * (lanes >> 5) + e] = BF6[Ac[:, e]] * sa
            Bm[32 * (lanes >> 5) + e, lanes & 31] = BF6[Bc[:, e]] * sb
        D = Am @ Bm
        C = st.regs(dfile)[d:d + 16].view(np.float32).astype(np.float64)
        out = np.zeros((16, 64), dtype=np.float32)
        for r in range(16):
            out[r] = (C[r] + D[_D32_ROWS[r], lanes & 31]).astype(np.float32)
        st.regs(dfile)[d:d + 16] = out.view(np.uint32)

    dd = vr(d, 16) if dfile == 'v' else ar(d, 16)
    rb = ar(b_agpr, 6) if bfile == 'a' else vr(b_agpr, 6)
    ra = ar(a, 6) if afile == 'a' else vr(a, 6)
    return Ins(text, 'mfma6', rd=ra + rb + dd + vr(scale_a) + vr(scale_b), wr=dd, emu=emu, tag=tag)


def mfma32_8(dfile, d, a, b_reg, scale_a, scale_b, tag='', bfile='a'):
    """D += A(e4m3 v[a:a+7], 32 x 64, E8M0 v[scale_a]) x B(e4m3 [bfile] b:b+7, 64 x 32, E8M0 v[scale_b]):
    v_mfma_scale_f32_32x32x64_f8f6f4 with both operands in OCP fp8 (cbsz:0 blgp:0): 64 matrix-pipe cycles.  Byte e of lane
    32h + r of A pairs with byte e of lane 32h + c of B (tools/fp8_probe.hip)."""
    rf = {'v': vreg, 'a': areg}
    text = ('v_mfma_scale_f32_32x32x64_f8f6f4 %s, %s, %s, %s, %s, %s op_sel_hi:[0,0,0]' %
            (rf[dfile](d, 16), vreg(a, 8), rf[bfile](b_reg, 8), rf[dfile](d, 16), vreg(scale_a), vreg(scale_b)))

    def emu(st):
        lanes = np.arange(64)
        Ac = st.V[a:a + 8].T.copy().view(np.uint8)            # [64, 32]
        Bc = st.regs(bfile)[b_reg:b_reg + 8].T.copy().view(np.uint8)
        sa = 2.0 ** (int(st.V[scale_a][0] & 0xff) - 127)
        sb = 2.0 ** (int(st.V[scale_b][0] & 0xff) - 127)
        Am = np.zeros((32, 64))
        Bm = np.zeros((64, 32))
        for e in range(32):
            Am[lanes & 31, 32 * (lanes >> 5) + e] = E4M3[Ac[:, e]] * sa
            Bm[32 * (lanes >> 5) + e, lanes & 31] = E4M3[Bc[:, e]] * sb
        D = Am @ Bm
        C = st.regs(dfile)[d:d + 16].view(np.float32).astype(np.float64)
        out = np.zeros((16, 64), dtype=np.float32)
        for r in range(16):
            out[r] = (C[r] + D[_D32_ROWS[r], lanes & 31]).astype(np.float32)
        st.regs(dfile)[d:d + 16] = out.view(np.uint32)

    dd = vr(d, 16) if dfile == 'v' else ar(d, 16)
    rb = ar(b_reg, 8) if bfile == 'a' else vr(b_reg, 8)
    return Ins(text, 'mfma6', rd=vr(a, 8) + rb + dd + vr(scale_a) + vr(scale_b), wr=dd, emu=emu, tag=tag)


def v_cvt_pk_fp8_f16(dst, dst_high, src_pk, scale_v):
    """half `dst_high` of v[dst] = the two e4m3 bytes of the packed f16 pair v[src_pk] / f32 v[scale_v] (low f16 -> low
    byte): v_cvt_scalef32_pk_fp8_f16 [op_sel:[0,0,1]]; writes 16 bits of its destination (dst-sel forwarding hazard)"""
    text = 'v_cvt_scalef32_pk_fp8_f16 %s, %s, %s%s' % (vreg(dst), vreg(src_pk), vreg(scale_v), ' op_sel:[0,0,1]' if dst_high else '')

    def emu(st):
        x = _halves(st.V[src_pk:src_pk + 1]).astype(np.float64)     # [64, 2]
        sc = st.f32('v', scale_v).astype(np.float64)[:, None]
        c = f_to_e4m3(x / sc).astype(np.uint32)
        w = c[:, 0] | (c[:, 1] << 8)
        if dst_high:
            st.V[dst] = (st.V[dst] & 0x0000ffff) | (w << 16)
        else:
            st.V[dst] = (st.V[dst] & 0xffff0000) | w
    return valu(text, vr(src_pk) + vr(scale_v), vr(dst), emu, partial=True)


# ---------------------------------------------------------------------------------------------
# list-scheduler items
# ---------------------------------------------------------------------------------------------
class Filler:
    __slots__ = ('ins', 'earliest', 'deadline', 'chain', 'seq', 'needs')

    def __init__(self, ins, earliest, deadline, chain, needs=()):
        self.ins = ins
        self.earliest = earliest  # may be issued after anchor #earliest has been emitted
        self.deadline = deadline  # must be issued before anchor #deadline
        self.chain = chain        # fillers of one chain keep their order
        self.seq = 0
        self.needs = needs        # tags of LDS operations that must have completed (counted lgkmcnt in front of it)



def f32_bits(x):
    return int(np.array([x], dtype=np.float32).view(np.uint32)[0])



def mfma_cycles(ins):
    """matrix-pipe cycles of an MFMA: 32 for the 32x32 shapes (64 with 8-bit operands on the block-scaled instruction: no
    cbsz / blgp modifier), 16 for the 16x16 ones"""
    if '32x32' in ins.text:
        return 64 if 'f8f6f4' in ins.text and 'cbsz' not in ins.text else 32
    return 16


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
        if i > 0 and stream[i - 1].kind == 'trans' and set(stream[i - 1].wr) & set(ins.rd):
            errs.append('%d: %s reads the result of the transcendental right before it' % (i, ins.text))
        if i > 0 and ins.kind == 'dma' and stream[i - 1].kind == 'salu' and ' m0,' in stream[i - 1].text:
            errs.append('%d: LDS-DMA right behind an M0 write' % i)
        if ins.kind in ('mfma16', 'mfma6'):
            dur = mfma_cycles(ins)
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
                if ins.kind in ('valu', 'trans') and r in mf_rd and t < mf_rd[r] + 12:  # (LDS data lands >= 64 cycles later)
                    errs.append('%d: %s overwrites %s%d read by an MFMA %d cycles ago' %
                                (i, ins.text, r[0], r[1], t - mf_rd[r]))
                if r in mf_end and t < mf_end[r]:
                    errs.append('%d: %s overwrites %s%d while an MFMA still writes it' % (i, ins.text, r[0], r[1]))
                if ins.kind in ('valu', 'trans'):
                    valu_wr[r] = i
                mf_end.pop(r, None)
            t += 4 * ins.cost if ins.kind in ('nop',) else (16 if ins.kind == 'trans' else 4)
    return errs


def model_cycles(body):
    """coarse issue model: cycles of one block"""
    t = pipe = 0
    for ins in body:
        if ins.kind in ('mfma16', 'mfma6'):
            st = max(t, pipe)
            pipe = st + mfma_cycles(ins)
            t = st + 8
        elif ins.kind == 'dma':
            t += 36
        elif ins.kind == 'trans':
            t += 16
        elif 'pk32_bf6' in ins.text:
            t += 32
        else:
            t += 4 * (ins.cost if ins.kind == 'nop' else 1)
    return max(t, pipe)

