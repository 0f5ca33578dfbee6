#!/usr/bin/env python3
"""Generator of the hand-scheduled gfx950 body loop of the R2L W256 ResMLP (fp16 main pass + two bf6 x bf6
correction terms, 32x32 MFMA shapes), plus a lane-accurate CPU emulator of the generated stream (isa.py).

What is computed (reference: model/nerf_raybased.py:443-465, ResMLP.forward, 43 blocks):
    x <- x + W2 relu(W1 x + b1) + b2            (activations in the act_scale domain)
with b2 folded on the host (x~_i = x_i - sum_{j<i} b2_j, b1'_i = b1_i + W1_i sum_{j<i} b2_j), so a
block is  h = relu(W1 x~ + b1'),  x~ += W2 h : the second layer accumulates IN PLACE into the fp32
residual stream, which is the MFMA C/D operand.

Arithmetic of one Linear(256,256):  y = hi(W) hi(a)              v_mfma_f32_32x32x16_f16      (1 pass)
                                      + bf6(W - hi(W)) bf6(a)    v_mfma_scale_f32_32x32x64_f8f6f4, e3m2 x e3m2,
                                      + bf6(W) bf6(a - hi(a))    4x the fp16 rate             (2 x 1/4 pass)
hi = fp16 rounding.  The correction terms need ~3 significant bits; OCP bf6 (e3m2) with one power-of-two scale
per layer and term (E8M0 operand of the instruction) keeps the network at L_inf ~2e-5 for uniform, Laplace,
sparse and outlier-laden weights (tools/quant_study.py); e2m3 weights or fp4 do not.

Why 32x32: a wave's in-order issue stalls on LDS-DMA, ds_read and the 32-wide conversions while the LDS array or the
VALU is busy; an MFMA shadows as many cycles of that as it lasts, and the 32x32 shapes last 32 cycles for the same
FLOPs per operand byte as two 16x16 ones (profiles/r02_cost_structure.txt: -10 % in a timing model of this loop).

Machine model (one wave64 = 32 rays = ONE column tile of 32, 4 waves per workgroup, one per SIMD; h = lane >> 5):
  VGPR   0..63   INh    fp16 B operands of layer 1 (= hi of x~): s*4, s = k-step of 16 features
        64..127  Hh     fp16 B operands of layer 2 (= hi of h)
       128..159  ACC    layer-1 accumulator of a row tile (32 features x 32 rays = 16 registers), 2 buffers (relu in place);
                        in layer 2, whose accumulator is X, the epilogue's copies of X
       160..175  BIAS   layer-1 bias of a row tile (C operand of its first MFMA)
       176..191  HI     fp16 weight fragments (A operand), 4 buffers
       192..203  A6     bf6 weight operands (A operand of the K=64 MFMA), 2 buffers x 6
       204..219  LO     fp16 pairs of a - hi(a) of the 2 row tiles being converted
       220..225  CV     conversion outputs
       226..     addresses / constants
  AGPR   0..127  X      fp32 residual stream = in-place accumulator of layer 2; X(u)+r = feature
                        32u + 8(r/4) + 4h + r%4 of ray lane & 31
       128..175  IN6    bf6 B operands of layer 1: a (t) 6 regs each | a - hi(a)
       176..223  H6     bf6 B operands of layer 2
The weight stream goes global -> LDS ring (4 slots x 28 KiB, LDS-DMA, 3 chunks ahead) -> ds_read.
A chunk = 1 row tile (32 output features) of one layer: 16 hi fragments (1 KiB) + 8 bf6 operands (1.5 KiB).
One counted vmcnt + one s_barrier per chunk (at its middle).

`schedule` interleaves the fixed MFMA anchor sequence with the filler instructions (LDS reads, epilogue VALU, LDS-DMA,
waits) by a small list scheduler.  `python body_gen.py --emit r2l_body_asm.inc` writes the inline-asm body; the tests
run the emulator against a float64 reference (tests/test_body_gen_cpu.py).
"""
import argparse
import sys

import numpy as np

from isa import (ACT_EXP, Ins, State, Filler, vr, ar, vreg, areg, sreg, mfma32_16, mfma32_6, ds_read_b128,
                 ds_read_b64, ds_read_b96, ds_max_u32, ds_write_b128, ds_write_b64, v_max3_abs, mfma32_8, v_cvt_pk_fp8_f16,
                 f_to_e4m3, v_mul_lit, v_fmac_lit, global_load_x4_a, ds_write_b128_stage, v_lshl_or, v_sub_imm, v_lshl_imm, waitcnt_lgkm, waitcnt_vm, barrier, valu, v_max0, v_accr, v_accw, v_cvt_pk_f16, v_resid16,
                 v_cvt_pk32_bf6, s_nop, salu, f_to_bf6, pack6, layer_exponent, weight_exps, f32_bits, check_hazards_stream,
                 model_cycles)

# ---------------------------------------------------------------------------------------------
# register map
# ---------------------------------------------------------------------------------------------
V_INH = 0
V_HH = 64
V_ACC = 128
V_BIAS = 160
# V_HI, V_A6, V_LO, V_CV (176 .. 225): configure()
V_L0 = 226        # lane*16                (LDS bytes 0 .. 65535)
V_L1 = 227        # lane*16 + 65536
V_L8A = 228       # lane*8                 (8-byte parts of the bf6 operands)
V_L8B = 229       # lane*8 + 65536
V_AUX = 230       # LDS aux base + (lane>>5)*16 (+4096 on odd blocks)
V_DMAOFF = 231    # wave*7168 + lane*16    (pieces 0..3; V_DMAOFF2 = +4096: pieces 4..6)
V_DMAOFF2 = 232
V_AUXOFF = 233    # wave*1024 + lane*16
V_LANE = 234
V_T = 235         # scratch of the exponent arithmetic
V_SC = 236        # 236,237: E8M0 scales (w - hi | w) of layer 1 of a block; 238,239: of layer 2
# activation exponents (one per operand set, calibrated per network: aux block): for the layer that CONSUMES a set the E8M0
# bytes of a and of a - hi(a), for the epilogue that PRODUCES it the f32 divisors 2^E and 2^(E-12) of the conversions
V_SB = 240        # 240,241: (a, a - hi(a)) of layer 1's source set | 242,243: of layer 2's
V_CVD = 244       # 244,245: divisors of the set layer 1 produces (H) | 246,247: of the set layer 2 produces (next IN)
V_ACT = 248       # 248..250: biased exponents 127 + E of (this block's IN set, its H set, the next block's IN set)
V_TAILB = 251     # LDS address of the tail table + (lane>>5)*16
V_BPERM = 252     # (lane ^ 32) * 4: ds_bpermute address of the other lane half
V_RAYOFF = 253    # lane * 12: byte offset of this lane's ray in the wave's rgb rows
V_RAY = 254
V_GMAX = 255      # range guard (opts.guard): running max |a| of the operand set being produced, per lane
N_VGPR_USED = 256
# NHI (fp16 fragment buffers): configure()

A_X = 0
A_IN6 = 128       # A_H6, N_AGPR_USED: configure()

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
S_RGB = 80      # 80,81 rgb rows of the launch's first tile (0: store the x image instead of the fused tail)
S_TAB = 82      # 82,83 tail table
S_NRAYS = 84
S_TILE0 = 85    # tile number of the launch's first tile within the call
S_T1 = 86       # temporaries 86..89
S_GPOS = 90     # range guard: LDS byte address of the maxima row (64 lanes x 4 B) of the operand set being produced
S_G2 = 92       # 92,93 staged transfer: S_G + 4096 (pieces 4..6 of a wave's share)
N_SGPR_LO, N_SGPR_HI = 40, 94
# Staged transfer of the weight stream (Opts.stage, bf6 only: its 32 free AGPRs): instead of LDS-DMA a wave loads its share of
# chunk T + 3 into 28 AGPRs (7 x global_load_dwordx4) at the rendezvous of row tile T and stores it into the ring at the next
# rendezvous (s_waitcnt vmcnt(0), then per piece ds_write_b128 from the AGPRs followed by the load of the piece of the chunk
# after); the barrier of a rendezvous certifies the stores of the one before.  An LDS-DMA instruction holds the wave's issue
# for 60 .. 185 cycles on this kernel (profiles/r02_cost_structure.txt: no LDS-DMA = -33 % time); a load + a 16-byte store
# hold it for ~20.
A_STG = 224

NSLOT = 4
AUX_BYTES = 4096           # per block: 256 f32 bias | 4 x (swl1, sw1, swl2, sw2) | pad
AUX_SCALES = 1024
AUX_ACT = 1088             # (127 + E_in, 127 + E_h, 127 + E_out, 0) as dwords, twice (one copy per lane half)
RES_SHIFT = 12             # a - hi(a) of an fp16-rounded value is converted 2^12 finer than the value
TAIL_BYTES = 4096
# range guard (the r2l_body_guard_kernel build of this stream): per operand set (2 per block: IN_b, H_b) one row of 32
# maxima of |a| (f32 bits; non-negative floats order as unsigned; lanes l and l + 32 share a word), accumulated over the
# tiles of a workgroup by ds_max_u32; the kernel's HIP epilogue reduces the rows and atomicMax-es them into the context's
# statistics
GSTAT_ROW = 128
# f16: the stream holds W x 2^F16_WSHIFT (hi and lo fragments alike; exact), so that lo(W) of weights ~ 2^-5 is a normal fp16
# number instead of a subnormal on the 2^-24 grid; the accumulators carry the factor, the epilogue takes it out (layer 1: one
# v_mul_f32 behind the relu; layer 2: X += acc / 2^F16_WSHIFT as one v_fmac_f32).  Layer 2 of this format accumulates W2 h in a
# fresh accumulator and joins X ONCE per row tile: 48 MFMAs rounding at ulp(X) each are what would dominate its error otherwise.
F16_WSHIFT = 8
V_XT = 216        # f16: 8 temporaries of the layer-2 epilogue (X values on their way through the VALU)
DEEP_RING = True  # bf6: fragment / operand rings extended into the free AGPRs (--shallow-ring: the round-2 rings of 4 / 2 buffers)
STAGE_ON = False  # set by --stage / Opts.stage users: the staging registers are the same AGPRs
DMA_POLICY = ''   # --dma-policy: cache-policy suffix of the stream's LDS-DMA instructions (' nt', ' sc1', ' sc0 sc1'); experiment
DMA6 = False      # --dma6 (bf6r): a wave's share as 6 x dwordx4, the sixth overlapping the fifth by 512 B, instead of 5 + 2 x dword

def set_ring(opts):
    """the deep rings and the staged transfer want the same free AGPRs of the bf6 build"""
    global DEEP_RING, STAGE_ON
    STAGE_ON = bool(getattr(opts, 'stage', False))
    DEEP_RING = not (STAGE_ON or getattr(opts, 'shallow_ring', False))


def configure(fmt):
    """Format of the two correction terms and source of the bf6(W) operand:
      'bf6'   OCP e3m2 x e3m2 (6 registers per K=64 operand, 32 matrix-pipe cycles), all 8 operands of a row tile streamed:
              28 KiB chunks (the round-2 kernel; kept as the reference build of the A/B);
      'bf6r'  the same arithmetic, but the four bf6(W) operands of a row tile (term 1: W x (a - hi(a))) are neither streamed
              nor read from LDS: every wave converts them from the fp16 fragments it holds for its own MFMAs (four
              v_cvt_scalef32_pk32_bf6_f16 per row tile, each in the shadow of a 32-cycle MFMA): 22 KiB chunks, -21 % of the
              L2 -> LDS stream AND of the LDS reads, no LDS traffic added (R2L_PREC_FP16_FP8 since round 3).  The fragment
              ring has 8 buffers (a K=64 step's four fragments are 16 consecutive registers), the layer-1 bias is read
              straight into the accumulator, the streamed bf6 operands go through AGPRs;
      'fp8'   OCP e4m3 x e4m3 (8 registers, 64 cycles; R2L_PREC_FP16_E4M3, one more mantissa bit in all four factors: half
              the error at 1.33x the MFMA time), all operands streamed: 32 KiB chunks.
    Sets the format-dependent part of the register map, the chunk geometry and the LDS map (module globals: the generators
    are run once per variant)."""
    global FMT, WREG, NA6, NHI, PIECES, CHUNK, SLOT, WAVE_BYTES, PW, LDS_AUX, LDS_TAIL, LDS_BYTES, LDS_GSTAT
    global V_HI, V_HL, V_A6, A_A6, V_LO, V_CV, V_WCV, V_WSC, V_DMAOFF4, A_H6, N_AGPR_USED, ORDER, ANCH_PER_TILE, A_INL, A_HLO
    global NHI_V, NA6B, A_FRX, A_A6X
    WREG = fmt == 'bf6r'
    FMT = 'bf6' if WREG else fmt
    ANCH_PER_TILE = 48 if FMT == 'f16' else 24      # MFMAs of a row tile
    NA6 = 6 if FMT == 'bf6' else 8                  # registers of a K=64 operand
    PIECES = 28 if FMT == 'bf6' else 32             # KiB of a chunk with all operands: 16 fp16 fragments + 8 x 1.5 | 2 KiB
                                                    # (f16: 16 hi + 16 lo fragments)
    CHUNK = 22 * 1024 if WREG else PIECES * 1024    # bytes of a chunk in the stream = LDS stride of the ring
    SLOT = CHUNK
    WAVE_BYTES = CHUNK // 4                         # a wave's share of a chunk: bf6r 5 x 1 KiB + 2 x 256 B
    PW = (6 if DMA6 else 7) if WREG else PIECES // 4   # LDS-DMA instructions per wave and chunk
    LDS_AUX = NSLOT * SLOT
    LDS_TAIL = LDS_AUX + 2 * AUX_BYTES              # tail table: 3 x 256 f32 (W_t / act_scale) | 2 x (3 folded biases, 0)
    LDS_BYTES = LDS_TAIL + TAIL_BYTES
    LDS_GSTAT = LDS_BYTES
    if WREG:
        NHI = 8                                     # fragment buffers: two groups of four = the sources of two conversions
        V_HI = 160                                  # 160..191 (the bias registers are gone: read straight into ACC)
        V_LO, V_CV, V_WCV, V_WSC, V_DMAOFF4 = 192, 208, 214, 220, 222
        V_A6, A_A6 = None, 224                      # streamed bf6 operands (term 0): 2 buffers in AGPRs
        ORDER = 'group'
    elif FMT == 'f16':
        # R2L_PREC_FP16X3 on this machine: three fp16 passes per k-step, hi(W) hi(a) + hi(W) lo(a) + lo(W) hi(a), lo = the fp16
        # rounding residual (exact products, no scales, no calibration: the correction terms are as good as fp16 allows).
        # The lo B operands live in the AGPRs the bf6 sets leave free; a second fragment ring holds lo(W).
        NHI = 4
        V_HI, V_HL, V_LO, V_CV = 176, 192, 208, 216
        V_A6, ORDER = None, 'f16'
        A_INL, A_HLO = 128, 192                     # lo(a) B operands of layer 1 / layer 2: + 4 s
    else:
        NHI = 4
        V_HI = 176
        V_A6 = 192                                  # weight operands of the K=64 MFMAs, 2 buffers
        V_LO = V_A6 + 2 * NA6                       # bf6: fp16 residual pairs of 2 row tiles (16); fp8: of one (8)
        V_CV = V_LO + (16 if FMT == 'bf6' else 8)   # conversion outputs: bf6 6; fp8 4 (values) + 4 (residuals)
        assert V_CV + (6 if FMT == 'bf6' else 8) <= V_L0
        ORDER = ORDER_BASE
    A_H6 = A_IN6 + 8 * NA6
    NHI_V, NA6B = NHI, 2                            # fragment buffers in VGPRs; buffers of the streamed K=64 operands
    if FMT == 'bf6' and not WREG and DEEP_RING:
        # a[224:255] are free in this build: 4 more fragment buffers and 2 more K=64 operand buffers.  Without them a read can be
        # issued at most 3 MFMAs (96 cycles) ahead of its use -- less than the LDS latency under the LDS-DMA writes: the counted
        # lgkmcnt waits were 14 % of the kernel (knock-out without them: 8.43 against 9.79 ms)
        NHI, NA6B = 8, 4
        A_FRX, A_A6X = A_H6 + 8 * NA6, A_H6 + 8 * NA6 + 16
        assert A_A6X + 2 * NA6 <= 256
    N_AGPR_USED = 256 if FMT == 'f16' else A_H6 + 8 * NA6 + (12 if WREG else 0)
    assert N_AGPR_USED <= 256


TILES = 16                 # row tiles of a block: layer * 8 + u
BIAS_AT = 17               # bf6r: anchor of a tile from which the NEXT tile's bias may be read into its accumulator buffer


def INH(s):
    return V_INH + s * 4


def HH(s):
    return V_HH + s * 4


def ACC(p):
    return V_ACC + p * 16


def HI(b):
    return V_HI + b * 4


def HIF(b):
    """buffer b of the fp16 fragment ring: (file, first register).  The bf6 build has 32 AGPRs left: buffers 4.. live there
    (ds_read and the MFMA's A operand take AGPRs), so that a fragment read can run up to NHI - 1 MFMAs ahead of its use"""
    return ('v', V_HI + b * 4) if b < NHI_V else ('a', A_FRX + (b - NHI_V) * 4)


def A6(b):
    """buffer b of the streamed K=64 weight operands: (file, first register)"""
    if WREG:
        return ('a', A_A6 + b * NA6)
    return ('v', V_A6 + b * NA6) if b < 2 else ('a', A_A6X + (b - 2) * NA6)


def X(u):
    return A_X + u * 16


def B6(base, term, t):
    return base + term * 4 * NA6 + t * NA6


# order of the eight K=64 MFMAs of a row tile: (term, t); term 0 = (w - hi) x bf6(a), 1 = w x bf6(a - hi)
J_ORDER = [(0, 0), (1, 0), (0, 1), (1, 1), (0, 2), (1, 2), (0, 3), (1, 3)]


# ---------------------------------------------------------------------------------------------
# layout maps shared with the host packer (r2l_common.h restated; tests compare both sides)
# ---------------------------------------------------------------------------------------------
def kappa(s, h, j):
    """input feature multiplied by element j of lane half h of fp16 k-step s (r2l_kappa32): the epilogue packs
    accumulator registers 4g .. 4g+3 (features 32u + 8g + 4h + i) of row tile u into k-step 2u + (g >> 1)"""
    return 32 * (s >> 1) + 16 * (s & 1) + 8 * (j >> 2) + 4 * h + (j & 3)


def mix_feat(t, h, e):
    """input feature multiplied by element e (0..31) of lane half h of K=64 step t: the conversion takes the 16 fp16
    pair registers of k-steps 4t .. 4t+3 in order"""
    return kappa(4 * t + (e >> 3), h, e & 7)


def off_hi(s):
    """byte offset inside a chunk (stream and ring slot alike) of the fp16 fragment of k-step s (1 KiB, lane * 16)"""
    return s * 1024


def off_a6(j):
    """(offset of the first 16 B/lane, offset of the rest) of K=64 operand j = (term, t) = J_ORDER[j] inside its ring slot.
    bf6: the rest is 8 B/lane (512 B); fp8: another 16 B/lane.  bf6r: only the term-0 operands exist in the stream; they
    follow the fragments back to back, 1,536 B each (1 KiB of lane * 16, then 512 B of lane * 8)."""
    term, t = J_ORDER[j]
    if WREG:
        assert term == 0
        base = 16 * 1024 + t * 1536
        return base, base + 1024
    if FMT == 'bf6':
        return (16 + j) * 1024, (24 + (j >> 1)) * 1024 + (j & 1) * 512
    return (16 + 2 * j) * 1024, (17 + 2 * j) * 1024


def off_lo(s):
    """f16: byte offset inside a chunk of the lo(W) fragment of k-step s"""
    return (16 + s) * 1024


def in_stream(j):
    """is operand j part of the weight stream (False: made on chip, bf6r term 1)"""
    return not (WREG and J_ORDER[j][0] == 1)


def pack_body_image(W1s, b1s, W2s, b2s, act_scale=16.0, act=None, fmt='bf6'):
    """Python restatement of the host packer (r2l_capi.hip pack_body_v3): returns (stream bytes,
    aux uint32 [n_block, 1024], total folded bias float64 [256]).  W*: [256, 256] float32 (out, in).
    act: 2 n_block + 1 activation exponents (IN set of block 0, H set of block 0, IN set of block 1, ...); default ACT_EXP"""
    configure(fmt)
    n_block = len(W1s)
    if act is None:
        act = [ACT_EXP] * (2 * n_block + 1)
    img = np.zeros(n_block * 16 * CHUNK, dtype=np.uint8)
    aux = np.zeros((n_block, AUX_BYTES // 4), dtype=np.uint32)
    Bsum = np.zeros(256, dtype=np.float64)
    lanes = np.arange(64)
    h = lanes >> 5
    r = lanes & 31
    for b in range(n_block):
        b1f = b1s[b].astype(np.float64) + W1s[b].astype(np.float64) @ Bsum
        aux[b, :256] = (b1f * act_scale * (2.0 ** F16_WSHIFT if FMT == 'f16' else 1.0)).astype(np.float32).view(np.uint32)
        for half in range(2):
            for i in range(3):
                aux[b, AUX_ACT // 4 + 4 * half + i] = 127 + act[2 * b + i]
        for layer, Wl in enumerate((W1s[b], W2s[b])):
            Wl = Wl.astype(np.float32)
            ex = layer_exponent(Wl)
            if FMT == 'f16':
                Wl = Wl * np.float32(2.0 ** F16_WSHIFT)
            hi = Wl.astype(np.float16)
            el, ew = weight_exps(ex, FMT)
            for qq in range(4):
                aux[b, AUX_SCALES // 4 + 4 * qq + 2 * layer] = 0x01010101 * (127 + el)
                aux[b, AUX_SCALES // 4 + 4 * qq + 2 * layer + 1] = 0x01010101 * (127 + ew)
            for u in range(8):
                base = ((b * 2 + layer) * 8 + u) * CHUNK
                rows = 32 * u + r
                for s in range(16):
                    p = base + off_hi(s)
                    frag = np.zeros((64, 8), dtype=np.float16)
                    for j in range(8):
                        frag[:, j] = hi[rows, kappa(s, h, j)]
                    img[p:p + 1024] = frag.view(np.uint8).reshape(-1)
                if FMT == 'f16':
                    lo16 = (Wl.astype(np.float64) - hi.astype(np.float64)).astype(np.float16)     # fp16(w - hi(w))
                    for s in range(16):
                        frag = np.zeros((64, 8), dtype=np.float16)
                        for j in range(8):
                            frag[:, j] = lo16[rows, kappa(s, h, j)]
                        img[base + off_lo(s):base + off_lo(s) + 1024] = frag.view(np.uint8).reshape(-1)
                    continue
                for j, (term, t) in enumerate(J_ORDER):
                    if not in_stream(j):
                        continue
                    codes = np.zeros((64, 32), dtype=np.uint8)
                    for e in range(32):
                        k = mix_feat(t, h, e)
                        w = Wl[rows, k].astype(np.float64)
                        if term == 0:
                            v = np.ldexp(w - hi[rows, k].astype(np.float64), -el)
                        else:
                            v = np.ldexp(w, -ew)
                        codes[:, e] = f_to_bf6(v) if FMT == 'bf6' else f_to_e4m3(v)
                    # bf6: 6 dwords per lane, little-endian 6-bit fields; fp8: byte e = element e, 8 dwords
                    words = pack6(codes) if FMT == 'bf6' else np.ascontiguousarray(codes).view(np.uint32)
                    o1, o2 = off_a6(j)
                    img[base + o1:base + o1 + 1024] = np.ascontiguousarray(words[:, :4]).view(np.uint8).reshape(-1)
                    rest = np.ascontiguousarray(words[:, 4:]).view(np.uint8).reshape(-1)
                    img[base + o2:base + o2 + rest.size] = rest
        Bsum = Bsum + b2s[b].astype(np.float64)
    return img, aux, Bsum


# ---------------------------------------------------------------------------------------------
# LDS-DMA of this kernel's ring
# ---------------------------------------------------------------------------------------------
def dma_piece(i, tag=''):
    """piece i of this wave's share of a chunk: global_load_lds_dwordx4 v_off, s[S_G:S_G+1] offset:imm with LDS destination
    M0 + imm + lane*16; pieces 4.. use the second offset register and M0 + 4096.  bf6r: a wave moves 5,632 B = pieces 0..4
    of 1 KiB and pieces 5, 6 of 256 B (global_load_lds_dword: 4 bytes per lane; dwordx3 does not pack, tools/dma12_probe.hip)."""
    if WREG and DMA6 and i == 5:
        voffr, width, imm = V_DMAOFF2, 16, 512                         # overlaps piece 4 by 512 B: the same bytes twice
    elif WREG and i >= 5:
        voffr, width, imm = V_DMAOFF4, 4, 1024 + 256 * (i - 5)        # relative to M0 + 4096 / the + 4096 offset register
    else:
        voffr, width, imm = (V_DMAOFF if i < 4 else V_DMAOFF2), 16, 1024 * (i & 3)
    text = 'global_load_lds_dword%s %s, %s offset:%d%s' % ('x4' if width == 16 else '', vreg(voffr), sreg(S_G, 2), imm, DMA_POLICY)

    def emu(st):
        copies = []
        g = st.S[S_G]
        for w in range(4):
            dw = (w - st.wave) * WAVE_BYTES
            for l in range(64):
                src = g + int(st.V[voffr][l]) + dw + imm
                dst = st.m0 + dw + imm + l * width
                assert 0 <= dst and dst + width <= LDS_AUX, dst
                assert 0 <= src and src + width <= len(st.img), src
                copies.append((dst, st.img[src:src + width].copy()))
                st.lds_pending[dst:dst + width] = True
        st.pend_dma.append(copies)
    return Ins(text, 'dma', rd=vr(voffr), emu=emu, cost=8 if width == 16 else 4, tag=tag)


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
                assert LDS_AUX <= dst and dst + 16 <= LDS_TAIL, dst
                copies.append((dst, st.aux[src:src + 16].copy()))
                st.lds_pending[dst:dst + 16] = True
        st.pend_dma.append(copies)
    return Ins(text, 'dma', rd=vr(V_AUXOFF), emu=emu, cost=8)


def stage_g2():
    """S_G2 = S_G + 4096: source of pieces 4..6 (the load's immediate reaches +-4 KiB)"""
    return [salu('s_add_u32 %s, %s, 0x1000' % (sreg(S_G2), sreg(S_G)), lambda st: st.S.__setitem__(S_G2, st.S[S_G] + 4096)),
            salu('s_addc_u32 %s, %s, 0' % (sreg(S_G2 + 1), sreg(S_G + 1)))]


def stage_load(i, tag=''):
    """piece i of this wave's share of the chunk at S_G -> staging AGPRs"""
    return global_load_x4_a(A_STG + 4 * i, V_DMAOFF, S_G if i < 4 else S_G2, 1024 * (i & 3), tag=tag)


def stage_write(i, slot, tag=''):
    """staging AGPRs of piece i -> ring slot `slot` (V_DMAOFF2 = V_DMAOFF + 64 KiB reaches slot 3)"""
    off = slot * SLOT + 1024 * i
    base, off = (V_DMAOFF, off) if off < 65536 else (V_DMAOFF2, off - 65536)
    return ds_write_b128_stage(base, A_STG + 4 * i, off, WAVE_BYTES, tag=tag)


def stage_prologue_ops():
    """ring prologue of the staged transfer: chunks 0, 1 in slots 0, 1 for everybody, chunk 2 on its way to the staging
    registers (the state every rendezvous leaves behind)"""
    def next_chunk():
        return [salu('s_add_u32 %s, %s, %s' % (sreg(S_G), sreg(S_W), sreg(S_POS)),
                     lambda st: st.S.__setitem__(S_G, st.S[S_W] + st.S[S_POS])),
                salu('s_addc_u32 %s, %s, 0' % (sreg(S_G + 1), sreg(S_W + 1))),
                salu('s_add_u32 %s, %s, 0x%x' % (sreg(S_POS), sreg(S_POS), CHUNK),
                     lambda st: st.S.__setitem__(S_POS, st.S[S_POS] + CHUNK)),
                salu('s_cmp_eq_u32 %s, %s' % (sreg(S_POS), sreg(S_END))),
                salu('s_cselect_b32 %s, 0, %s' % (sreg(S_POS), sreg(S_POS)),
                     lambda st: st.S.__setitem__(S_POS, 0 if st.S[S_POS] == st.S[S_END] else st.S[S_POS]))] + stage_g2()
    ops = []
    for c in range(3):
        ops += next_chunk() + [stage_load(i) for i in range(PW)]
        if c < 2:
            ops += [waitcnt_vm(0)] + [stage_write(i, c) for i in range(PW)]
    return ops + [waitcnt_lgkm(0), barrier()]


# ---------------------------------------------------------------------------------------------
# block schedule
# ---------------------------------------------------------------------------------------------
def slot_of(T):
    return T % NSLOT


def lds_addr(slot, byte_off, width):
    """(base VGPR, immediate offset) of byte `byte_off` of ring slot `slot` for a per-lane width of 16 or 8 bytes"""
    off = slot * SLOT + byte_off
    lo, hi = (V_L0, V_L1) if width == 16 else (V_L8A, V_L8B)
    return (lo, off) if off < 65536 else (hi, off - 65536)


ORDER_BASE = 'tail'   # 'tail': the 16 fp16 MFMAs of a row tile, then its 8 K=64 MFMAs; 'mix': the K=64 ones behind k-steps
                      # 8..15; bf6r always runs 'group': per K=64 step t its four fp16 MFMAs, then (1, t), then (0, t)
ORDER = ORDER_BASE


def tile_anchors():
    """the 24 MFMAs of a row tile as (kind, s or j), all on ONE accumulator (an accumulate chain).  The B operands of the
    previous layer's last row tile are converted late: its k-steps 14, 15 and its K=64 operands (., 3) come last."""
    out = []
    if ORDER == 'f16':       # three fp16 passes per k-step on one accumulate chain
        for s in range(16):
            out += [('m16', s), ('mhl', s), ('mlh', s)]
        return out
    if ORDER == 'group':     # bf6r: K=64 step t = fp16 k-steps 4t .. 4t+3, then term 1 (operand converted from them), then term 0
        for t in range(4):
            out += [('m16', 4 * t + i) for i in range(4)]
            out += [('m6', J_ORDER.index((1, t))), ('m6', J_ORDER.index((0, t)))]
        return out
    for s in range(16):
        out.append(('m16', s))
        if ORDER == 'mix' and s >= 8:
            out.append(('m6', s - 8))
    if ORDER == 'tail':
        out += [('m6', j) for j in range(8)]
    return out


# ANCH_PER_TILE (MFMAs of a row tile: 24, f16: 48): configure()
_POS = {}


def anchor_index(T, kind, sj):
    """global anchor number of an MFMA (T may be <0 or >=16: neighbouring block iterations)"""
    if _POS.get('order') != ORDER:
        _POS.clear()
        _POS['order'] = ORDER
        for i, key in enumerate(tile_anchors()):
            _POS[key] = i
    return T * ANCH_PER_TILE + _POS[(kind, sj)]


def guard_flush(tag):
    """range guard: the per-lane maximum of the operand set just completed joins its LDS row; next set, maximum cleared.
    The row has 32 words: lanes l and l + 32 share one (V_BPERM & 0x7c = 4 (lane & 31))."""
    return [valu('v_and_b32 %s, 0x7c, %s' % (vreg(V_T), vreg(V_BPERM)), vr(V_BPERM), vr(V_T),
                 lambda st: st.V.__setitem__(V_T, st.V[V_BPERM] & np.uint32(0x7c))),
            valu('v_add_u32 %s, %s, %s' % (vreg(V_T), sreg(S_GPOS), vreg(V_T)), vr(V_T), vr(V_T),
                 lambda st: st.V.__setitem__(V_T, (st.V[V_T] + np.uint32(st.S[S_GPOS])).astype(np.uint32))),
            ds_max_u32(V_T, V_GMAX, 0, tag=tag),       # S_GPOS carries LDS_GSTAT (beyond a 16-bit offset)
            salu('s_add_u32 %s, %s, 0x%x' % (sreg(S_GPOS), sreg(S_GPOS), GSTAT_ROW),
                 lambda st: st.S.__setitem__(S_GPOS, st.S[S_GPOS] + GSTAT_ROW)),
            valu('v_mov_b32 %s, 0' % vreg(V_GMAX), (), vr(V_GMAX), lambda st: st.V.__setitem__(V_GMAX, np.zeros(64, np.uint32)))]


def epilogue_ops(T, guard=False, split=False):
    """VALU epilogue of row tile T (block-local tile index, may be -1 = tile 15 of the previous block).  Returns
    [(Ins, consumer)]; consumer: None | ('hi', s) | ('b6', term, t).  Works in the tile's own accumulator registers:
    relu in place (layer 1), copies of X (layer 2, whose accumulator is X itself).  guard: the values also enter the
    running maximum of their operand set (2 v_max3_f32 per 4 values)."""
    Tm = T % TILES
    layer, u = Tm >> 3, Tm & 7
    acc = ACC(T & 1)
    hset, b6 = (HH, A_H6) if layer == 0 else (INH, A_IN6)
    ops = []
    for g in range(4):
        t = [acc + 4 * g + i for i in range(4)]
        if FMT == 'f16' and layer == 1:
            # x <- x + acc / 2^F16_WSHIFT through temporaries (split: the ray tile's initial split, nothing to add yet)
            a_ = t
            t = [V_XT + 4 * (g & 1) + i for i in range(4)]
            for i in range(4):
                ops.append((v_accr(t[i], X(u) + 4 * g + i), None))
            for i in range(4):
                if not split:
                    ops.append((v_fmac_lit(t[i], 2.0 ** -F16_WSHIFT, a_[i]), None))
            for i in range(4):
                if not split:
                    ops.append((v_accw(X(u) + 4 * g + i, t[i]), None))
        else:
            for i in range(4):
                ops.append((v_max0(t[i], t[i]) if layer == 0 else v_accr(t[i], X(u) + 4 * g + i), None))
            if FMT == 'f16':       # the accumulator carries the weight stream's factor
                for i in range(4):
                    ops.append((v_mul_lit(t[i], t[i], 2.0 ** -F16_WSHIFT), None))
        if guard:
            ops.append((v_max3_abs(V_GMAX, t[0], t[1]), None))
            ops.append((v_max3_abs(V_GMAX, t[2], t[3]), None))
        s = 2 * u + (g >> 1)
        h01 = hset(s) + 2 * (g & 1)
        h23 = h01 + 1
        lo = V_LO + (u & 1) * 8 + g * 2
        ops.append((v_cvt_pk_f16(h01, t[0], t[1]), ('hi', s)))
        ops.append((v_cvt_pk_f16(h23, t[2], t[3]), ('hi', s)))
        # half-register writes: low halves first, then the high halves (never two writers of one register back to back)
        if FMT in ('fp8', 'f16'):
            lo = V_LO + g * 2
        ops.append((v_resid16(lo, 0, h01, 0, t[0], S_NEG1), None))
        ops.append((v_resid16(lo + 1, 0, h23, 0, t[2], S_NEG1), None))
        ops.append((v_resid16(lo, 1, h01, 1, t[1], S_NEG1), None))
        ops.append((v_resid16(lo + 1, 1, h23, 1, t[3], S_NEG1), None))
        if FMT == 'f16':   # the residual pairs ARE the lo B operand of k-step s: registers 2 (g & 1), + 1
            lset = A_HLO if layer == 0 else A_INL
            ops.append((v_accw(lset + 4 * s + 2 * (g & 1), lo), ('lo', s)))
            ops.append((v_accw(lset + 4 * s + 2 * (g & 1) + 1, lo + 1), ('lo', s)))
        if FMT == 'fp8':
            # four values = one register of each e4m3 operand (byte e = 16 (u & 1) + 4 g + i of K=64 step u >> 1): two
            # half-register conversions each, never two writers of one register back to back
            cvh, cvl, tt, r = V_CV + g, V_CV + 4 + g, u >> 1, 4 * (u & 1) + g
            ops.append((v_cvt_pk_fp8_f16(cvh, 0, h01, V_CVD + 2 * layer), None))
            ops.append((v_cvt_pk_fp8_f16(cvl, 0, lo, V_CVD + 2 * layer + 1), None))
            ops.append((v_cvt_pk_fp8_f16(cvh, 1, h23, V_CVD + 2 * layer), None))
            ops.append((v_cvt_pk_fp8_f16(cvl, 1, lo + 1, V_CVD + 2 * layer + 1), None))
            ops.append((v_accw(B6(b6, 0, tt) + r, cvh), ('b6', 0, tt)))
            ops.append((v_accw(B6(b6, 1, tt) + r, cvl), ('b6', 1, tt)))
    if FMT == 'bf6' and (u & 1) == 1:
        tt = u >> 1
        # 32-wide conversions of the finished pair of row tiles; the independent one first (dst-sel forwarding)
        ops.append((v_cvt_pk32_bf6(V_CV, hset(4 * tt), V_CVD + 2 * layer), None))
        for i in range(6):
            ops.append((v_accw(B6(b6, 0, tt) + i, V_CV + i), ('b6', 0, tt)))
        ops.append((v_cvt_pk32_bf6(V_CV, V_LO, V_CVD + 2 * layer + 1), None))
        for i in range(6):
            ops.append((v_accw(B6(b6, 1, tt) + i, V_CV + i), ('b6', 1, tt)))
    return ops


def derive_sb(dst, act):
    """v[dst], v[dst+1] = E8M0 bytes (replicated) of an operand set and of its fp16 residuals, from its biased exponent"""
    return [v_lshl_or(V_T, act, 8, act), v_lshl_or(dst, V_T, 16, V_T), v_sub_imm(dst + 1, dst, 0x01010101 * RES_SHIFT)]


def derive_cv(dst, act):
    """v[dst], v[dst+1] = f32 divisors 2^E and 2^(E-12) of the two conversions that produce an operand set"""
    return [v_lshl_imm(dst, 23, act), v_sub_imm(dst + 1, dst, RES_SHIFT << 23)]


def derive_wsc(dst, sc):
    """v[dst] = 2^(E - 127) as f32 from the replicated E8M0 byte in v[sc] (bf6r: divisor of the bf6(W) conversion)"""
    return [valu('v_and_b32 %s, 0xff, %s' % (vreg(V_T), vreg(sc)), vr(sc), vr(V_T),
                 lambda st: st.V.__setitem__(V_T, st.V[sc] & np.uint32(0xff))),
            v_lshl_imm(dst, 23, V_T)]


def read_act(tag):
    return ds_read_b96(V_ACT, V_AUX, AUX_ACT, tag=tag)


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

    def need(self, it, key, also=()):
        """make sure the LDS read registered under `key` has landed; reads in `also` that have been issued by now ride along
        (one s_waitcnt for a run of MFMAs: the instruction itself holds the wave's issue for some cycles, needed or not)"""
        if key not in self.ds_index and it == 0:
            return  # issued by the iteration before the schedule starts (iteration 0 is never the extracted one)
        idx = self.ds_index[key]
        if idx < self.ds_done:
            return
        for k2 in also:
            if k2 in self.ds_index:
                idx = max(idx, self.ds_index[k2])
        n = min(15, self.ds_issued - idx - 1)         # the counter has 4 bits; waiting for more is always right
        self.emit(it, waitcnt_lgkm(n))
        self.ds_done = self.ds_issued - n


def build_fillers(it, opts):
    """fillers of block iteration `it` (anchors numbered it*384 + ...)"""
    F = []
    base_anchor = it * TILES * ANCH_PER_TILE

    def A(T, kind, sj):
        return base_anchor + anchor_index(T, kind, sj)

    for T in range(TILES):
        layer, u = T >> 3, T & 7
        slot = slot_of(T)
        # --- weight operand reads: one ds_read_b128 per fp16 k-step (one MFMA each) into a ring of NHI buffers, two reads
        # per bf6 operand into two buffers; every read runs `rd_lead` anchors ahead of its MFMA at most
        if layer == 0 and not WREG:
            for g in range(4):
                F.append(Filler(ds_read_b128(V_BIAS + 4 * g, V_AUX, 128 * u + 32 * g, tag=('bias', it, T, g)),
                                A(T - 1, 'm16', 0), A(T, 'm16', 0), ('rd',)))
        if layer == 0 and WREG:
            # no bias registers: the bias is read straight into the tile's accumulator buffer, which the epilogue of tile
            # T - 2 (same buffer) has left by anchor BIAS_AT of tile T - 1 (its deadline is pulled in accordingly)
            for g in range(4):
                F.append(Filler(ds_read_b128(ACC(T & 1) + 4 * g, V_AUX, 128 * u + 32 * g, tag=('bias', it, T, g)),
                                A(T - 1, 'm16', 0) + BIAS_AT, A(T, 'm16', 0), ('rd',)))
        def cvt_window(TT, t):
            """bf6r: (anchor behind which conversion t of tile TT may issue = its predecessor's MFMA has read the operand
            register, anchor before which it must issue = two instructions ahead of its own MFMA)"""
            pm = A(TT, 'm6', J_ORDER.index((1, t - 1))) if t else A(TT - 1, 'm6', J_ORDER.index((1, 3)))
            if TT % TILES == 0 and t == 0:
                pm = max(pm, A(TT, 'm16', 0) - 1)      # the block's divisor is derived at the loop head
            return pm + 1, A(TT, 'm16', 4 * t + 2)

        for s_ in range(16):
            n = T * 16 + s_
            prev = n - NHI                                 # last user of the buffer
            earliest = max(A(prev // 16, 'm16', prev % 16), A(T, 'm16', s_) - opts.rd_lead)
            deadline = A(T, 'm16', s_)
            if FMT == 'f16':   # hi(W) of k-step s feeds two MFMAs; a second ring streams lo(W)
                earliest = max(A(prev // 16, 'mhl', prev % 16), A(T, 'm16', s_) - 3 * opts.rd_lead)
                bv, off = lds_addr(slot, off_lo(s_), 16)
                F.append(Filler(ds_read_b128(V_HL + 4 * (n % NHI), bv, off, tag=('hl', it, T, s_)),
                                max(A(prev // 16, 'mlh', prev % 16), A(T, 'mlh', s_) - 3 * opts.rd_lead), A(T, 'mlh', s_), ('rdl',)))
            if WREG:
                # the buffer is also a source of the conversion of its group (prev's K=64 step), and the whole group of this
                # fragment must be on its way when its own conversion may issue
                TP, tp = prev // 16, (prev % 16) >> 2
                earliest = max(A(TP, 'm16', prev % 16), cvt_window(TP, tp)[1], A(T, 'm16', s_) - opts.rd_lead - 4)
                deadline = min(deadline, cvt_window(T, s_ >> 2)[0])
            bv, off = lds_addr(slot, off_hi(s_), 16)
            fl, rg = HIF(n % NHI)
            F.append(Filler(ds_read_b128(rg, bv, off, tag=('hi', it, T, s_), dfile=fl), earliest, deadline, ('rd',)))
        for j in range(8):
            if not in_stream(j) or FMT == 'f16':
                continue
            n = T * 4 + (j >> 1) if WREG else T * 8 + j
            prevj = [jj for jj in range(8) if in_stream(jj)]
            k = prevj.index(j)
            # the buffer's last user: the operand NA6B before this one in stream order
            if k >= NA6B:
                pa = A(T, 'm6', prevj[k - NA6B])
            else:
                pa = A(T - 1, 'm6', prevj[len(prevj) - NA6B + k])
            earliest = max(pa, A(T, 'm6', j) - opts.rd_lead6)
            deadline = A(T, 'm6', j)
            o1, o2 = off_a6(j)
            fl, reg = A6(n % NA6B)
            bv, off = lds_addr(slot, o1, 16)
            F.append(Filler(ds_read_b128(reg, bv, off, tag=('a6', it, T, j, 0), dfile=fl), earliest, deadline, ('rd6',)))
            if FMT == 'bf6':
                bv, off = lds_addr(slot, o2, 8)
                F.append(Filler(ds_read_b64(reg + 4, bv, off, tag=('a6', it, T, j, 1), dfile=fl), earliest, deadline, ('rd6',)))
            else:
                bv, off = lds_addr(slot, o2, 16)
                F.append(Filler(ds_read_b128(reg + 4, bv, off, tag=('a6', it, T, j, 1)), earliest, deadline, ('rd6',)))
        if WREG:
            # bf6(W / 2^(e-4)) of K=64 step t from the four fragments the wave holds for its own fp16 MFMAs: after the last of
            # them has landed and the previous conversion's MFMA has read the operand register; two instructions before its MFMA
            for t in range(4):
                e, d = cvt_window(T, t)
                if 'wcvt' in getattr(opts, 'drop', ()):       # diagnostics (wrong results): what the conversions cost
                    continue
                F.append(Filler(v_cvt_pk32_bf6(V_WCV, HI(4 * (t & 1)), V_WSC + layer), e, d, ('wcv',), needs=(('hi', it, T, 4 * t + 3),)))
        if T == 8:
            # this block's layer-2 scales were read during layer 1; flip to the next block's aux slot, then fetch
            # the next block's layer-1 scales (the running layer 1 is over: its scale registers are free)
            F.append(Filler(valu('v_xor_b32 %s, 0x%x, %s' % (vreg(V_AUX), AUX_BYTES, vreg(V_AUX)), vr(V_AUX), vr(V_AUX),
                                 lambda st: st.V.__setitem__(V_AUX, st.V[V_AUX] ^ AUX_BYTES)),
                            A(T, 'm16', 1), A(T, 'm16', 12), ('auxflip',)))
            if FMT != 'f16':
                F.append(Filler(ds_read_b64(V_SC, V_AUX, AUX_SCALES, tag=('scale', it + 1, 0)),
                                A(T, 'm16', 1), A(T + 1, 'm16', 0), ('auxflip',)))
                # ... and its activation exponents (this block's were used up at tile 1)
                F.append(Filler(read_act(('act', it + 1)), A(T, 'm16', 1), A(T + 1, 'm16', 0), ('auxflip',)))
        if T == 1 and FMT != 'f16':
            # layer-2 scales of this block (layer 2 of the previous block is over)
            F.append(Filler(ds_read_b64(V_SC + 2, V_AUX, AUX_SCALES + 8, tag=('scale', it, 1)),
                            A(T, 'm16', 1), A(T + 1, 'm16', 0), ('auxflip',)))
        # --- epilogue of the PREVIOUS tile, under this tile's MFMAs ----------------------------
        Tprev = T - 1
        pu = (Tprev % TILES) & 7
        nl_T0 = (Tprev - pu) + 8  # first tile of the consuming layer, block-local
        e0 = A(T, 'm16', 0) + 2   # two further MFMAs behind the last writer of the accumulator
        epi = epilogue_ops(Tprev, opts.guard)
        if opts.guard and pu == 7:    # the last row tile of a layer: its operand set is complete
            epi += [(ins, None) for ins in guard_flush(('gmax', it, T))]
        for ins, cons in epi:
            dl = A(T + 1, 'm16', 0)  # latest: the accumulator buffer is reused by tile T+1
            if WREG and ((T + 1) % TILES) < 8:
                dl = A(T, 'm16', 0) + BIAS_AT     # ... whose bias is read into it from this anchor on
            if cons is not None:
                if cons[0] == 'hi':
                    first = A(nl_T0, 'm16', cons[1])
                elif cons[0] == 'lo':
                    first = A(nl_T0, 'mhl', cons[1])
                else:
                    first = A(nl_T0, 'm6', J_ORDER.index((cons[1], cons[2])))
                dl = min(dl, first - 2)
            F.append(Filler(ins, e0, dl, ('epi',)))
        # --- rendezvous + refill at the middle of each chunk (= row tile) ------------------------
        a0 = base_anchor + T * ANCH_PER_TILE + (opts.rdv_at if opts.rdv_at is not None else ANCH_PER_TILE // 3)
        ch = ('dma',)
        F.append(Filler(waitcnt_vm(0 if opts.stage else PW), a0 - 1, a0 + 1, ch))
        F.append(Filler(barrier(), a0 - 1, a0 + 1, ch))
        seq = []
        tgt_slot = (T + 3) % NSLOT
        seq.append(salu('s_add_u32 %s, %s, %s' % (sreg(S_G), sreg(S_W), sreg(S_POS)),
                        lambda st: st.S.__setitem__(S_G, st.S[S_W] + st.S[S_POS])))
        seq.append(salu('s_addc_u32 %s, %s, 0' % (sreg(S_G + 1), sreg(S_W + 1))))
        seq.append(salu('s_add_u32 %s, %s, 0x%x' % (sreg(S_POS), sreg(S_POS), CHUNK),
                        lambda st: st.S.__setitem__(S_POS, st.S[S_POS] + CHUNK)))
        seq.append(salu('s_cmp_eq_u32 %s, %s' % (sreg(S_POS), sreg(S_END))))
        seq.append(salu('s_cselect_b32 %s, 0, %s' % (sreg(S_POS), sreg(S_POS)),
                        lambda st: st.S.__setitem__(S_POS, 0 if st.S[S_POS] == st.S[S_END] else st.S[S_POS])))
        if T == 0:
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
        if opts.stage:
            seq += stage_g2()
            for i in range(PW):       # the chunk that arrived -> slot of chunk T + 2, then the same registers take chunk T + 3
                seq.append(stage_write(i, (T + 2) % NSLOT, tag=('stw', it, T, i)))
                seq.append(stage_load(i, tag=('dma', it, T, i)))
        else:
            seq.append(salu('s_mov_b32 m0, %s' % sreg(S_M0SLOT + tgt_slot),
                            lambda st, k=S_M0SLOT + tgt_slot: setattr(st, 'm0', st.S[k])))
            seq.append(s_nop(0))
            for i in range(PW):
                if i == 4:
                    seq.append(salu('s_add_u32 m0, m0, 0x1000', lambda st: setattr(st, 'm0', st.m0 + 4096)))
                    seq.append(s_nop(0))
                seq.append(dma_piece(i, tag=('dma', it, T, i)))
        end = a0 + ANCH_PER_TILE - ANCH_PER_TILE // 12      # well before the next rendezvous
        if opts.dma_burst:
            for ins in seq:
                F.append(Filler(ins, a0 - 1, a0 + 1, ch))
        else:
            # SALU prelude right behind the barrier, then the pieces spread over the following MFMAs
            # 4 waves x 7 pieces of 1 KiB right behind the barrier ask the 64 B/clk path into the CU for 128 B/clk; at one piece
            # per two to four MFMAs the waves wait less (bf6 -0.7 %, e4m3 -1.2 %, f16 -2.0 % kernel time:
            # profiles/r03_dma_experiments.txt item 6)
            gap = opts.dma_gap if opts.dma_gap is not None else (1 if WREG or opts.stage else {'bf6': 3, 'fp8': 2, 'f16': 4}[FMT])
            first_piece = next(i for i, x in enumerate(seq) if x.kind in ('dma', 'vload', 'ds') and x.tag)
            for ins in seq[:first_piece]:
                F.append(Filler(ins, a0 - 1, a0 + 3, ch))
            k = 0
            for ins in seq[first_piece:]:
                F.append(Filler(ins, a0 + 1 + gap * k, end, ch))
                if ins.kind in ('dma', 'vload'):
                    k += 1
    return F


def schedule(opts, n_iter=3):
    """list-schedule n_iter block iterations; returns [(iteration_of_position, Ins)] where the
    position's iteration is that of the surrounding anchors"""
    set_ring(opts)
    configure(opts.fmt)
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
    deep = FMT == 'bf6' and not WREG and DEEP_RING
    # with the deep rings the reads run far enough ahead for one wait to cover a short run of MFMAs (-1 % on top of the rings' -1 %)
    wg16 = opts.wait_group if opts.wait_group is not None else (3 if deep else 2 if FMT in ('fp8', 'f16') else 1)
    wg6 = opts.wait_group6 if opts.wait_group6 is not None else (2 if deep else 1)
    per_block = TILES * ANCH_PER_TILE
    total_anchors = n_iter * per_block
    anchors = tile_anchors()

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
        for key in f.needs:
            sch.need(it, key)
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
        it = a // per_block
        T = (a // ANCH_PER_TILE) % TILES
        kind, sj = anchors[a % ANCH_PER_TILE]
        while True:  # forced fillers: deadline reached
            r = [f for f in ready(a - 1) if f.deadline <= a]
            if not r:
                break
            issue(r[0], it)
        layer, u = T >> 3, T & 7
        hset = INH if layer == 0 else HH
        if a % per_block == 0:
            # loop head: the prefetch issued by the previous iteration's tail -- or by the prologue, which lacks the
            # tail's other reads, so a counted wait would be too generous there -- is drained completely
            sch.emit(it, waitcnt_lgkm(0))
            sch.ds_done = sch.ds_issued
            if WREG:   # f32 divisor 2^(e-4) of this block's layer-1 conversions, from the E8M0 byte its MFMAs use
                for ins in derive_wsc(V_WSC, V_SC + 1):
                    sch.emit(it, ins)
        if WREG and a % ANCH_PER_TILE == 0 and T == 2:
            sch.need(it, ('scale', it, 1))          # ... and of its layer-2 conversions (tiles 8..15; scales read during tile 1)
            for ins in derive_wsc(V_WSC + 1, V_SC + 3):
                sch.emit(it, ins)
        if FMT != 'f16' and a % ANCH_PER_TILE == 0 and T == 1:
            # layer 2 of the previous block and its last epilogue are over: the exponents of this block's H set (layer 2
            # consumes it) and of the set layer 2 produces
            for ins in derive_sb(V_SB + 2, V_ACT + 1) + derive_cv(V_CVD + 2, V_ACT + 2):
                sch.emit(it, ins)
        if FMT != 'f16' and a % ANCH_PER_TILE == 0 and T == 9:
            # layer 1 and its last epilogue are over: the next block's exponents (read behind the aux flip of tile 8)
            sch.need(it, ('act', it + 1))
            for ins in derive_sb(V_SB, V_ACT) + derive_cv(V_CVD, V_ACT + 1):
                sch.emit(it, ins)
        dfile, d = ('v', ACC(T & 1)) if layer == 0 or FMT == 'f16' else ('a', X(u))
        if kind in ('mhl', 'mlh'):      # f16: hi(W) x lo(a) | lo(W) x hi(a)
            n = T * 16 + sj
            if kind == 'mhl':
                lset = A_INL if layer == 0 else A_HLO
                ins = mfma32_16(dfile, d, HI(n % NHI), lset + 4 * sj, dfile, d, tag=(kind, it, T, sj), bfile='a')
            else:
                sch.need(it, ('hl', it, T, sj))
                ins = mfma32_16(dfile, d, V_HL + 4 * (n % NHI), hset(sj), dfile, d, tag=(kind, it, T, sj))
            cap = opts.cap16
        elif kind == 'm16':
            sch.need(it, ('hi', it, T, sj), [('hi', it, T, sj + g) for g in range(1, wg16) if sj + g < 16])
            n = T * 16 + sj
            if layer == 0 and sj == 0:
                sch.need(it, ('bias', it, T, 3))
                fl, rg = HIF(n % NHI)
                ins = mfma32_16('v', d, rg, hset(sj), 'v', d if WREG else V_BIAS, tag=('m16', it, T, sj), afile=fl)
            elif FMT == 'f16' and sj == 0:      # layer 2: a fresh accumulator (its bias is folded into later layer-1 biases)
                ins = mfma32_16('v', d, HI(n % NHI), hset(sj), '0', 0, tag=('m16', it, T, sj))
            else:
                fl, rg = HIF(n % NHI)
                ins = mfma32_16(dfile, d, rg, hset(sj), dfile, d, tag=('m16', it, T, sj), afile=fl)
            cap = opts.cap16
        else:
            term, t = J_ORDER[sj]
            n = T * 4 + t if WREG else T * 8 + sj
            if in_stream(sj):
                sch.need(it, ('a6', it, T, sj, 1), [('a6', it, T, sj + g, 1) for g in range(1, wg6) if sj + g < 8])
            if u == 0 and (kind, sj) == [x for x in anchors if x[0] == 'm6'][0]:
                sch.need(it, ('scale', it, layer))
            if sj == 0 and ORDER == 'tail' and opts.chain_nop >= 0:
                sch.emit(it, s_nop(opts.chain_nop))      # fp16 -> scaled MFMA on one accumulator: keep them apart
            b6 = A_IN6 if layer == 0 else A_H6
            if FMT == 'fp8':
                ins = mfma32_8(dfile, d, A6(n % NA6B)[1], B6(b6, term, t), V_SC + 2 * layer + term, V_SB + 2 * layer + term,
                               tag=('m6', it, T, sj))
            else:
                fl, reg = ('v', V_WCV) if not in_stream(sj) else A6(n % NA6B)     # bf6r term 1: the operand just converted
                ins = mfma32_6(dfile, d, reg, B6(b6, term, t), V_SC + 2 * layer + term, V_SB + 2 * layer + term,
                               tag=('m6', it, T, sj), afile=fl)
            cap = opts.cap6 * (1 if FMT == 'bf6' else 2)    # a 64-cycle MFMA shadows twice the issue slots
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
        self.rd_lead = 4          # anchors (MFMAs of 32 cycles) an fp16 fragment read is issued ahead of its MFMA, at most
        self.rd_lead6 = 4
        self.cap16 = 6            # issue slots for fillers behind an MFMA
        self.cap6 = 6
        self.dma_gap = None       # anchors between two LDS-DMA pieces (None: 3 for the bf6 stream, else 1)
        self.dma_burst = False
        self.chain_nop = -1       # s_nop N between the last fp16 and the first K=64 MFMA of a row tile (-1: none)
        self.skip_terms = ()      # diagnostics: drop the K=64 MFMAs of these correction terms (wrong results)
        self.guard = False        # the range-guard build: per operand set the maximum |a| over every ray of the launch
        self.fmt = 'bf6'          # correction terms: 'bf6' (e3m2 x e3m2) | 'fp8' (e4m3 x e4m3): configure()
        self.stage = False        # weight stream through staging AGPRs + ds_write_b128 instead of LDS-DMA (bf6 only)
        self.wait_group = None    # fp16 fragments one s_waitcnt may cover (those already issued); None: 3 with the deep rings, else 1
        self.wait_group6 = None   # ... K=64 operands; None: 2 with the deep rings, else 1
        self.rdv_at = None        # anchor of a row tile in front of which its rendezvous sits (None: a third into the tile)
        self.shallow_ring = False # bf6: fragment / operand rings of 4 / 2 buffers in VGPRs only (the round-2 kernel)
        self.__dict__.update(kw)


def act_prologue():
    """exponent registers in front of the first tile: layer 1 consumes IN(0) and produces H(0); the initial split of X
    produces IN(0) with the divisors of the layer-2 epilogue"""
    if FMT == 'f16':
        return []
    return ([read_act(('act', 0)), waitcnt_lgkm(0)] + derive_sb(V_SB, V_ACT) + derive_cv(V_CVD, V_ACT + 1) +
            derive_cv(V_CVD + 2, V_ACT))


def steady_block(opts):
    """(prologue reads, loop body) -- the body is iteration 1 of a 3-iteration schedule; the prologue is
    the set of LDS reads of iteration 1 that the schedule placed inside iteration 0 (re-issued before
    the loop is entered), in their issue order."""
    out = schedule(opts, 3)
    body = [ins for it, ins in out if it == 1]
    pro = [ins for it, ins in out if it == 0 and ins.kind == 'ds' and ins.tag[0] != 'stw' and ins.tag[1] == 1]
    return pro, body


# ---------------------------------------------------------------------------------------------
# whole-kernel text
# ---------------------------------------------------------------------------------------------
def split_ops(u, guard=False):
    """standalone split of X row tile u -> the layer-1 operand sets (the layer-2 epilogue without MFMAs)"""
    return [ins for ins, _ in epilogue_ops(8 + u, guard, split=True)]


def f16_join_ops(u, p):
    """f16: X row tile u += accumulator buffer p / 2^F16_WSHIFT (the join of a layer-2 row tile, outside the loop schedule)"""
    ops = []
    for i in range(16):
        ops += [v_accr(V_XT + (i & 7), X(u) + i), v_fmac_lit(V_XT + (i & 7), 2.0 ** -F16_WSHIFT, ACC(p) + i), v_accw(X(u) + i, V_XT + (i & 7))]
    return ops


def f16_clear_ops():
    """f16: the loop head joins "tile 15 of the previous block" into X row tile 7: nothing in front of the first block"""
    return [valu('v_mov_b32 %s, 0' % vreg(ACC(1) + i), (), vr(ACC(1) + i),
                 (lambda i_: lambda st: st.V.__setitem__(ACC(1) + i_, np.zeros(64, np.uint32)))(i)) for i in range(16)]


def fused_tail_text():
    """rgb = sigmoid(W_t (x + h) + b_t) of the wave's 32 rays, straight from the residual stream in the AGPRs
    (model/nerf_raybased.py:539-544 with the global skip of :541; the standalone r2l_tail_kernel computes the same from the
    stored images).  h, the head output, is read back from the image operand %3 points to (the block loop consumed xin); a lane sums its 128 features
    x 3 channels in fp32, ds_bpermute adds the two lane halves, lanes 0..31 store 12 bytes each.  Registers: everything
    below V_L0 is dead between the last block and the next tile's split."""
    L = []
    a = L.append
    XA, WB, TMP, ACCS, RGBR, BIASR, TR = 0, 128, 176, 180, 186, 190, 194
    v_lane, v_rayoff, v_ray = V_LANE, V_RAYOFF, V_RAY
    NBUF = 4                                  # groups of weights in flight: 12 ds_read_b128 <= the 4-bit lgkmcnt
    # h comes through operand %3: with the fused tail nothing is written to xout, so the operand carries the HEAD image instead --
    # the same address as xin when the launch runs all blocks, another one when it continues a residual stream that an earlier
    # launch left in xin (R2L_PREC_FP16_SPLIT: blocks [0, split) by the bf6 kernel, the rest by this one; round 5)
    a('s_add_u32 %s, %s, %s' % (sreg(S_T0 + 4), sreg(S_XOUT), sreg(S_TILEOFF)))
    a('s_addc_u32 %s, %s, %s' % (sreg(S_T0 + 5), sreg(S_XOUT + 1), sreg(S_TILEOFF + 1)))
    for i in range(32):
        a('global_load_dwordx4 %s, %s, %s offset:%d' % (vreg(XA + 4 * i, 4), vreg(V_L0), sreg(S_T0 + 4, 2), (i % 4) * 1024))
        if i % 4 == 3:
            a('s_add_u32 %s, %s, 0x1000' % (sreg(S_T0 + 4), sreg(S_T0 + 4)))
            a('s_addc_u32 %s, %s, 0' % (sreg(S_T0 + 5), sreg(S_T0 + 5)))

    def rd_w(i):
        for c in range(3):   # features 8 i + 4 h .. + 3 of channel c
            a('ds_read_b128 %s, %s offset:%d' % (vreg(WB + 12 * (i % NBUF) + 4 * c, 4), vreg(V_TAILB), 1024 * c + 32 * i))

    for i in range(6):
        a('v_mov_b32 %s, 0' % vreg(ACCS + i))
    for i in range(NBUF):
        rd_w(i)
    for i in range(32):
        for k in range(4):
            a('v_accvgpr_read_b32 %s, %s' % (vreg(TMP + k), areg(A_X + 4 * i + k)))
        a('s_waitcnt vmcnt(%d)' % (31 - i))
        for k in range(4):
            a('v_add_f32 %s, %s, %s' % (vreg(XA + 4 * i + k), vreg(XA + 4 * i + k), vreg(TMP + k)))
        a('s_waitcnt lgkmcnt(%d)' % (3 * (min(i + NBUF - 1, 31) - i)))
        for k in range(4):
            for c in range(3):
                acc = vreg(ACCS + 2 * c + (k & 1))
                a('v_fma_f32 %s, %s, %s, %s' % (acc, vreg(XA + 4 * i + k), vreg(WB + 12 * (i % NBUF) + 4 * c + k), acc))
        if i + NBUF < 32:
            rd_w(i + NBUF)
    a('ds_read_b128 %s, %s offset:3072' % (vreg(BIASR, 4), vreg(V_TAILB)))
    for c in range(3):
        a('v_add_f32 %s, %s, %s' % (vreg(ACCS + 2 * c), vreg(ACCS + 2 * c), vreg(ACCS + 2 * c + 1)))
    for c in range(3):
        a('ds_bpermute_b32 %s, %s, %s' % (vreg(TR + c), vreg(V_BPERM), vreg(ACCS + 2 * c)))
    a('s_waitcnt lgkmcnt(0)')
    for c in range(3):
        a('v_add_f32 %s, %s, %s' % (vreg(TR + c), vreg(TR + c), vreg(ACCS + 2 * c)))
        a('v_add_f32 %s, %s, %s' % (vreg(TR + c), vreg(TR + c), vreg(BIASR + c)))
        a('v_mul_f32 %s, 0xbfb8aa3b, %s' % (vreg(TR + c), vreg(TR + c)))      # -log2(e)
    for c in range(3):
        a('v_exp_f32 %s, %s' % (vreg(TR + c), vreg(TR + c)))
    a('s_nop 1')
    for c in range(3):
        a('v_add_f32 %s, 1.0, %s' % (vreg(TR + c), vreg(TR + c)))
    for c in range(3):
        a('v_rcp_f32 %s, %s' % (vreg(RGBR + c), vreg(TR + c)))
    a('s_nop 1')
    # first ray of this wave: ((tile0 + tile) * 4 + wave) * 32; rows of 12 bytes
    a('s_add_u32 %s, %s, %s' % (sreg(S_T1), sreg(S_TILE), sreg(S_TILE0)))
    a('s_lshl_b32 %s, %s, 2' % (sreg(S_T1), sreg(S_T1)))
    a('s_add_u32 %s, %s, %s' % (sreg(S_T1), sreg(S_T1), sreg(S_WAVE)))
    a('s_lshl_b32 %s, %s, 5' % (sreg(S_T1), sreg(S_T1)))
    a('v_add_u32 %s, %s, %s' % (vreg(v_ray), sreg(S_T1), vreg(v_lane)))
    a('s_mul_hi_u32 %s, %s, 12' % (sreg(S_T1 + 1), sreg(S_T1)))
    a('s_mul_i32 %s, %s, 12' % (sreg(S_T1), sreg(S_T1)))
    a('s_add_u32 %s, %s, %s' % (sreg(S_T1), sreg(S_T1), sreg(S_RGB)))
    a('s_addc_u32 %s, %s, %s' % (sreg(S_T1 + 1), sreg(S_T1 + 1), sreg(S_RGB + 1)))
    a('v_cmp_gt_u32 vcc, %s, %s' % (sreg(S_NRAYS), vreg(v_ray)))
    a('s_mov_b32 %s, -1' % sreg(S_T1 + 2))
    a('s_mov_b32 %s, 0' % sreg(S_T1 + 3))
    a('s_and_b64 exec, vcc, %s' % sreg(S_T1 + 2, 2))                      # lanes 0..31 with a ray inside the call
    a('global_store_dwordx3 %s, %s, %s' % (vreg(v_rayoff), vreg(RGBR, 3), sreg(S_T1, 2)))
    a('s_mov_b64 exec, -1')
    return L


def kernel_text(opts):
    """asm text of the whole body kernel (one inline-asm statement).  Inputs (asm operands):
    %0 wimg (s64)  %1 aux (s64)  %2 xin (s64)  %3 xout (s64)  %4 n_tiles  %5 n_block  %6 wave  %7 blockIdx.x
    %8 gridDim.x  %9 rgb (s64; 0: store the x image to xout, else the fused tail writes rgb and %3 is the head image h)
    %10 tail table (s64)  %11 n_rays  %12 number of the launch's first tile"""
    set_ring(opts)
    configure(opts.fmt)
    pro, body = steady_block(opts)
    L = []
    a = L.append
    a('s_mov_b32 %s, m0' % sreg(S_M0SAVE))
    if FMT == 'fp8':   # MODE.FP16_OVFL: conversions to e4m3 clamp at +-448 instead of producing NaN (tools/fp8_probe.hip)
        a('s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1')
    a('s_mov_b64 %s, %%0' % sreg(S_W, 2))
    a('s_mov_b64 %s, %%1' % sreg(S_AUXB, 2))
    a('s_mov_b64 %s, %%2' % sreg(S_XIN, 2))
    a('s_mov_b64 %s, %%3' % sreg(S_XOUT, 2))
    a('s_mov_b32 %s, %%4' % sreg(S_NTILES))
    a('s_mov_b32 %s, %%5' % sreg(S_NBLOCK))
    a('s_mov_b32 %s, %%6' % sreg(S_WAVE))
    a('s_mov_b32 %s, %%7' % sreg(S_TILE))
    a('s_mov_b32 %s, %%8' % sreg(S_GRID))
    a('s_mov_b64 %s, %%9' % sreg(S_RGB, 2))
    a('s_mov_b64 %s, %%10' % sreg(S_TAB, 2))
    a('s_mov_b32 %s, %%11' % sreg(S_NRAYS))
    a('s_mov_b32 %s, %%12' % sreg(S_TILE0))
    a('s_mov_b32 %s, 0xbf800000' % sreg(S_NEG1))
    # lane id, LDS / DMA offsets
    a('v_mbcnt_lo_u32_b32 %s, -1, 0' % vreg(V_LANE))
    a('v_mbcnt_hi_u32_b32 %s, -1, %s' % (vreg(V_LANE), vreg(V_LANE)))
    a('v_lshlrev_b32 %s, 4, %s' % (vreg(V_L0), vreg(V_LANE)))
    a('v_add_u32 %s, 0x10000, %s' % (vreg(V_L1), vreg(V_L0)))
    a('v_lshlrev_b32 %s, 3, %s' % (vreg(V_L8A), vreg(V_LANE)))
    a('v_add_u32 %s, 0x10000, %s' % (vreg(V_L8B), vreg(V_L8A)))
    a('v_lshrrev_b32 %s, 5, %s' % (vreg(V_AUX), vreg(V_LANE)))
    a('v_lshlrev_b32 %s, 4, %s' % (vreg(V_AUX), vreg(V_AUX)))
    a('v_add_u32 %s, 0x%x, %s' % (vreg(V_TAILB), LDS_TAIL, vreg(V_AUX)))
    a('v_add_u32 %s, 0x%x, %s' % (vreg(V_AUX), LDS_AUX, vreg(V_AUX)))
    a('v_xor_b32 %s, 32, %s' % (vreg(V_BPERM), vreg(V_LANE)))
    a('v_lshlrev_b32 %s, 2, %s' % (vreg(V_BPERM), vreg(V_BPERM)))
    a('v_mul_u32_u24 %s, 12, %s' % (vreg(V_RAYOFF), vreg(V_LANE)))
    a('s_mul_i32 %s, %s, 0x%x' % (sreg(S_T0), sreg(S_WAVE), WAVE_BYTES))        # this wave's share of a chunk
    a('v_add_u32 %s, %s, %s' % (vreg(V_DMAOFF), sreg(S_T0), vreg(V_L0)))
    a('v_add_u32 %s, 0x%x, %s' % (vreg(V_DMAOFF2), 0x10000 if opts.stage else 0x1000, vreg(V_DMAOFF)))
    if WREG:   # pieces 5, 6 move 4 bytes per lane: wave * share + 4096 + lane * 4
        a('v_lshlrev_b32 %s, 2, %s' % (vreg(V_DMAOFF4), vreg(V_LANE)))
        a('v_add_u32 %s, %s, %s' % (vreg(V_DMAOFF4), sreg(S_T0), vreg(V_DMAOFF4)))
        a('v_add_u32 %s, 0x1000, %s' % (vreg(V_DMAOFF4), vreg(V_DMAOFF4)))
    a('s_lshl_b32 %s, %s, 10' % (sreg(S_T0 + 1), sreg(S_WAVE)))                 # wave * 1024
    a('v_add_u32 %s, %s, %s' % (vreg(V_AUXOFF), sreg(S_T0 + 1), vreg(V_L0)))
    for k in range(NSLOT):
        a('s_add_u32 %s, %s, 0x%x' % (sreg(S_M0SLOT + k), sreg(S_T0), k * SLOT))
    a('s_add_u32 %s, %s, 0x%x' % (sreg(S_AUXM0), sreg(S_T0 + 1), LDS_AUX))
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

    # tail table -> LDS (4 KiB, 1 KiB per wave; the oldest load of the kernel: landed with the first chunk)
    a('s_add_u32 m0, %s, 0x%x' % (sreg(S_T0 + 1), LDS_TAIL))
    a('s_nop 0')
    a('global_load_lds_dwordx4 %s, %s' % (vreg(V_AUXOFF), sreg(S_TAB, 2)))
    L += issue_aux
    if opts.stage:
        assert FMT == 'bf6' and not WREG and A_H6 + 8 * NA6 <= A_STG
        L += [ins.text for ins in stage_prologue_ops()]
    else:
        for k in range(3):
            L += issue_chunk(k)
        a('s_waitcnt vmcnt(%d)' % (2 * PW))
        a('s_barrier')
    # activation exponents of block 0 (later blocks and tiles: derived inside the block loop; the last block's "next IN
    # set" is block 0's, so the state at the end of a tile is the state the next tile starts from)
    for ins in act_prologue():
        a(ins.text)
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
        if 'xload' in getattr(opts, 'drop', ()):      # diagnostics (wrong results): what the exposed x load of a ray tile costs
            continue
        a('global_load_dwordx4 %s, %s, %s offset:%d' % (areg(A_X + 4 * i, 4), vreg(V_L0), sreg(S_T0 + 4, 2), (i % 4) * 1024))
        if i % 4 == 3:
            a('s_add_u32 %s, %s, 0x1000' % (sreg(S_T0 + 4), sreg(S_T0 + 4)))
            a('s_addc_u32 %s, %s, 0' % (sreg(S_T0 + 5), sreg(S_T0 + 5)))
    a('s_waitcnt vmcnt(0)')
    if 'xload2' in getattr(opts, 'drop', ()):      # diagnostics (same results): the x load a second time, exposed like the first
        a('s_add_u32 %s, %s, %s' % (sreg(S_T0 + 4), sreg(S_XIN), sreg(S_TILEOFF)))
        a('s_addc_u32 %s, %s, %s' % (sreg(S_T0 + 5), sreg(S_XIN + 1), sreg(S_TILEOFF + 1)))
        for i in range(32):
            a('global_load_dwordx4 %s, %s, %s offset:%d sc1' % (areg(A_X + 4 * i, 4), vreg(V_L0), sreg(S_T0 + 4, 2), (i % 4) * 1024))
            if i % 4 == 3:
                a('s_add_u32 %s, %s, 0x1000' % (sreg(S_T0 + 4), sreg(S_T0 + 4)))
                a('s_addc_u32 %s, %s, 0' % (sreg(S_T0 + 5), sreg(S_T0 + 5)))
        a('s_waitcnt vmcnt(0)')
    # initial split of row tiles 0..6 (tile 7's runs at the head of the loop body)
    if opts.guard:
        a('v_mov_b32 %s, 0' % vreg(V_GMAX))
        a('s_mov_b32 %s, 0x%x' % (sreg(S_GPOS), LDS_GSTAT))
    for u in range(7):
        for ins in split_ops(u, opts.guard):
            a(ins.text)
    if FMT == 'f16':
        for ins in f16_clear_ops():
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
    # drain the prefetch reads, let the last MFMAs retire
    a('s_waitcnt lgkmcnt(0)')
    a('s_nop 15')
    a('s_nop 15')
    if FMT == 'f16':      # the last row tile of the last block joins X here (inside the loop: at the head of the next block)
        for ins in f16_join_ops(7, 1):
            a(ins.text)
        a('s_nop 1')
    a('s_cmp_eq_u64 %s, 0' % sreg(S_RGB, 2))
    a('s_cbranch_scc1 L_storex_%=')
    L += fused_tail_text()
    a('s_branch L_next_%=')
    # ---- x image out (r2l_debug_body, networks without the global skip) ------------------------
    a('L_storex_%=:')
    a('s_add_u32 %s, %s, %s' % (sreg(S_T0 + 4), sreg(S_XOUT), sreg(S_TILEOFF)))
    a('s_addc_u32 %s, %s, %s' % (sreg(S_T0 + 5), sreg(S_XOUT + 1), sreg(S_TILEOFF + 1)))
    for i in range(32):
        a('global_store_dwordx4 %s, %s, %s offset:%d' % (vreg(V_L0), areg(A_X + 4 * i, 4), sreg(S_T0 + 4, 2), (i % 4) * 1024))
        if i % 4 == 3:
            a('s_add_u32 %s, %s, 0x1000' % (sreg(S_T0 + 4), sreg(S_T0 + 4)))
            a('s_addc_u32 %s, %s, 0' % (sreg(S_T0 + 5), sreg(S_T0 + 5)))
    a('L_next_%=:')
    a('s_add_u32 %s, %s, %s' % (sreg(S_TILE), sreg(S_TILE), sreg(S_GRID)))
    a('s_cmp_lt_u32 %s, %s' % (sreg(S_TILE), sreg(S_NTILES)))
    a('s_cbranch_scc1 L_tile_%=')
    a('L_exit_%=:')
    a('s_waitcnt vmcnt(0) lgkmcnt(0)')
    a('s_barrier')
    if FMT == 'fp8':
        a('s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 0')
    a('s_mov_b32 m0, %s' % sreg(S_M0SAVE))
    return L, pro, body


def emit_inc(path, opts):
    L, pro, body = kernel_text(opts)
    drop = getattr(opts, 'drop', ())   # diagnostics only (wrong results): timing knock-outs of instruction classes
    if drop:
        texts = set()
        for ins in body:
            gone = (ins.kind in drop or (ins.kind == 'wait' and 'lgkm' in drop and 'lgkmcnt' in ins.text) or
                    ('stw' in drop and ins.text.startswith('ds_write_b128')))
            if gone:
                texts.add(ins.text)
        keep_always = ('s_waitcnt vmcnt', 's_barrier')
        first = L.index('L_block_%=:')
        last = len(L) - 1 - L[::-1].index('s_sub_u32 %s, %s, 1' % (sreg(S_BLK), sreg(S_BLK)))
        L = L[:first + 1] + [t for t in L[first + 1:last] if not (t in texts and not t.startswith(keep_always))] + L[last:]
    if 'lgkmnop' in drop:      # every counted LDS wait replaced by an s_nop 0: same instruction count, no waiting (wrong results)
        first = L.index('L_block_%=:')
        L = L[:first + 1] + ['s_nop 0' if t.startswith('s_waitcnt lgkmcnt') else t for t in L[first + 1:]]
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
    set_ring(opts)
    configure(opts.fmt)
    pro, body = steady_block(opts)
    st = State(wave, img, aux, n_block, LDS_BYTES + (2 * n_block * GSTAT_ROW if opts.guard else 0))
    lanes = np.arange(64, dtype=np.uint32)
    st.V[V_LANE] = lanes
    st.V[V_L0] = lanes * 16
    st.V[V_L1] = lanes * 16 + 65536
    st.V[V_L8A] = lanes * 8
    st.V[V_L8B] = lanes * 8 + 65536
    st.V[V_AUX] = LDS_AUX + (lanes >> 5) * 16
    st.V[V_DMAOFF] = wave * WAVE_BYTES + lanes * 16
    st.V[V_DMAOFF2] = wave * WAVE_BYTES + (65536 if opts.stage else 4096) + lanes * 16
    if WREG:
        st.V[V_DMAOFF4] = wave * WAVE_BYTES + 4096 + lanes * 4
    st.V[V_AUXOFF] = wave * 1024 + lanes * 16
    S = st.S
    S[S_W] = 0
    S[S_AUXB] = 0
    S[S_POS] = 0
    S[S_AUXPOS] = 0
    S[S_END] = n_block * 16 * CHUNK
    S[S_AUXEND] = n_block * AUX_BYTES
    for k in range(NSLOT):
        S[S_M0SLOT + k] = wave * WAVE_BYTES + k * SLOT
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
    if opts.stage:
        st.run(stage_prologue_ops())
    else:
        for k in range(3):
            issue_chunk(k)
        waitcnt_vm(2 * PW).emu(st)
        barrier().emu(st)
    st.run(act_prologue())
    st.A[A_X:A_X + 128] = np.ascontiguousarray(x_tile_regs, dtype=np.float32).view(np.uint32)
    st.V[V_BPERM] = (lanes ^ 32) * 4
    S[S_GPOS] = LDS_GSTAT
    for u in range(7):
        st.run(split_ops(u, opts.guard))
    if FMT == 'f16':
        st.run(f16_clear_ops())
    st.run(pro)
    for b in range(n_block):
        st.run(body)
    waitcnt_lgkm(0).emu(st)
    if FMT == 'f16':
        st.run(f16_join_ops(7, 1))
    errs = list(st.errors)
    if check_hazards:
        errs += check_hazards_stream(body + body)
    if opts.guard:   # the maxima rows (f32) of the 2 n_block operand sets ride along
        return st.A[A_X:A_X + 128].view(np.float32).copy(), errs, st.lds[LDS_GSTAT:LDS_GSTAT + 2 * n_block * GSTAT_ROW].view(np.float32).reshape(2 * n_block, 32).copy()
    return st.A[A_X:A_X + 128].view(np.float32).copy(), errs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--emit', help='write the inline-asm include file')
    ap.add_argument('--dma-burst', action='store_true')
    ap.add_argument('--rd-lead', type=int, default=4)
    ap.add_argument('--rd-lead6', type=int, default=4)
    ap.add_argument('--cap16', type=int, default=6)
    ap.add_argument('--cap6', type=int, default=6)
    ap.add_argument('--dma-gap', type=int, default=None)
    ap.add_argument('--chain-nop', type=int, default=-1)
    ap.add_argument('--order', default=None, choices=['tail', 'mix'])
    ap.add_argument('--guard', action='store_true', help='the range-guard build of the stream (r2l_body_guard_kernel)')
    ap.add_argument('--fmt', default='bf6', choices=['bf6', 'bf6r', 'fp8', 'f16'],
                    help='correction terms: bf6 (e3m2, all operands streamed) | bf6r (bf6(W) converted from the fp16 fragments in registers) | fp8 (e4m3) | '
                         'f16 (three fp16 passes: R2L_PREC_FP16X3 on this machine)')
    ap.add_argument('--wait-group', type=int, default=None)
    ap.add_argument('--wait-group6', type=int, default=None)
    ap.add_argument('--rdv-at', type=int, default=None)
    ap.add_argument('--dma-policy', default='')
    ap.add_argument('--shallow-ring', action='store_true', help='bf6: the round-2 rings (4 fragment / 2 operand buffers, VGPRs only)')
    ap.add_argument('--stage', action='store_true', help='bf6: the weight stream through 28 staging AGPRs + ds_write_b128 instead of LDS-DMA')
    ap.add_argument('--dma6', action='store_true', help='bf6r: 6 x dwordx4 per wave and chunk (512 B moved twice) instead of 5 x dwordx4 + 2 x dword')
    ap.add_argument('--dump', help='write the loop body as plain text')
    ap.add_argument('--skip-terms', default='', help='diagnostics only: comma list of correction terms to drop')
    ap.add_argument('--drop', default='', help='diagnostics only: comma list of instruction classes left out of the block loop '
                    '(lgkm, dma, valu, ds, mfma6, mfma16, wcvt): timing knock-outs, wrong results')
    a = ap.parse_args()
    if a.order:
        global ORDER_BASE
        ORDER_BASE = a.order
    if a.dma_policy:
        global DMA_POLICY
        DMA_POLICY = ' ' + a.dma_policy.replace('+', ' ')
    if a.dma6:
        global DMA6
        DMA6 = True
    opts = Opts(dma_burst=a.dma_burst, rd_lead=a.rd_lead, rd_lead6=a.rd_lead6, cap16=a.cap16, cap6=a.cap6, dma_gap=a.dma_gap,
                chain_nop=a.chain_nop, guard=a.guard, fmt=a.fmt, stage=a.stage, shallow_ring=a.shallow_ring, rdv_at=a.rdv_at, wait_group=a.wait_group, wait_group6=a.wait_group6, skip_terms=tuple(int(t) for t in a.skip_terms.split(',') if t),
                drop=tuple(x for x in a.drop.split(',') if x))
    if a.emit:
        n = emit_inc(a.emit, opts)
        print('wrote', a.emit, n)
    if a.dump:
        pro, body = steady_block(opts)
        with open(a.dump, 'w') as f:
            for ins in body:
                f.write(ins.text + '\n')
        print('model cycles per block', model_cycles(body))


configure('bf6')

if __name__ == '__main__':
    sys.exit(main())
