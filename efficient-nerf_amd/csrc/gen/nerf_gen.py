#!/usr/bin/env python3
"""Generator of the hand-scheduled gfx950 layer chain of the NeRF teacher MLP (model/nerf_raybased.py:377-401,
NeRF.forward with use_viewdirs, 8 x 256 + view branch): the 16x16-shape instance of the machine of isa.py (body_gen.py is the 32x32 one).

One straight-line asm block computes all eleven Linear layers for the 128 points of a workgroup tile (a wave owns 32
points = two column tiles of 16), from the embedding fragments the HIP prologue leaves in AGPRs to raw = (rgb, sigma) in
8 VGPRs:

  layer  source (B operands)                     MFMAs per row tile and column tile      row tiles  epilogue     destination
  L0     pts embedding E (AGPR, fp16 hi | lo)    2 k-steps x 3 passes                    16         relu         Q
  L1     Q                                       8 fp16 + 4 bf6 (K = 128)                16         relu         P   (L2: P -> Q, L3, L4)
  L5     E, then Q                               2 x 3, 8 + 4                            16         relu         P
  L6, L7 P -> Q -> P
  FA     P -> feature (no activation) | alpha    8 + 4                                   17         - | sigma    Q | out
  V      view embedding (AGPR), then Q           1 x 3, 8 + 4                            8          relu         P (k-steps 0..3)
  RGB    P (128 wide)                            4 + 2                                   1          rgb          out

Arithmetic of a 256- (128-) wide source as in the R2L body (body_gen.py): fp16 main pass + bf6(W - hi(W)) x bf6(a) + bf6(W) x bf6(a - hi(a))
at 4x the fp16 rate.  The embedding k-steps (sines and cosines: not reducible to 3 bits) run as three fp16 passes on hi / lo
fragments of both operands and come FIRST in a row tile, so a layer's first MFMAs do not wait for the previous layer's last
epilogue.  Weights are unscaled; every layer carries its two E8M0 scale bytes next to its bias.

LDS: 4 ring slots of 32 KiB + 16 KiB resident bias / scale table (loaded once per workgroup by the HIP prologue).  The
weight stream of one tile is 80 chunks, each a whole number of row tiles (L0 2 x 8 row tiles, standard layers 8 x 2, L5
16 x 1, FA 9 x 2, V 4 x 2, RGB 1); 80 = 0 mod 4 and every tile starts at stream offset 0, so every LDS and stream address
is an immediate.  Ring protocol as in the R2L body (3 chunks ahead, one counted vmcnt wait + barrier per chunk, LDS-DMA).
A tile block starts with `s_waitcnt vmcnt(0)` + barrier (chunks 0..2 were issued by the previous block's tail, or by the
kernel prologue) and ends with the next tile's chunks 0..2 in flight.

`python nerf_gen.py --emit DIR` writes nerf_mlp_asm.inc (the tile block; placeholders %[eh00] ... for the 12 AGPR inputs,
%[o0] ... %[o7] for the outputs, %[wimg] %[wave]) and nerf_mlp_pro_asm.inc (the ring prologue).  tests/test_nerf_gen_cpu.py
runs the lane-accurate emulator against a float64 evaluation of the network and the static hazard check of isa.py.
"""
import argparse
import os
import sys

import numpy as np

import isa as B
from isa import (Ins, vr, ar, vreg, areg, sreg, waitcnt_lgkm, waitcnt_vm, barrier, valu, v_max0, v_accw, v_accr, v_cvt_pk_f16,
                 v_cvt_pk32_bf6, s_nop, salu, ds_read_b128, ds_read_b64, f_to_bf6, pack6, layer_exponent, weight_exps,
                 f32_bits, Filler, check_hazards_stream, kappa16 as kappa, mix16 as mix_feat)

# ---------------------------------------------------------------------------------------------
# register map
# ---------------------------------------------------------------------------------------------
V_SET = {'P': 0, 'Q': 64}     # fp16 hi B operands of the two activation sets: + c*32 + s*4
V_ACC, V_BIAS, V_HI, V_A6, V_LO, V_TMP = 128, 144, 152, 168, 180, 212
V_L0, V_L1, V_L8A, V_L8B, V_AUX, V_LANE = 232, 233, 234, 235, 236, 237
V_SBA, V_SBL, V_CVA, V_CVL = 238, 239, 240, 241
V_SC = 242                    # 242,243 | 244,245: E8M0 weight scales of even | odd layers
V_LOFF = 246                  # lane*16 + 4096 (second LDS-DMA group)
N_VGPR_CLOBBER = 248          # v248.. belong to the compiler (the 8 outputs)
A_SET = {'P': 0, 'Q': 48}     # bf6 B operands: + term*24 + (t*2+c)*6
A_E = 96                      # emulator numbering of the inputs: Eh 96 + (e*2+c)*4, El 112 + ..., Vh 128 + c*4, Vl 136 + c*4
N_AGPR_CLOBBER = 96

S_W = 40                      # 40,41 stream base
S_WAVE = 42
S_NEG1, S_M0SAVE = 43, 44
S_G = 46                      # 46,47 LDS-DMA source
S_WPW = 48                    # 48..55: wave * PW * 1024 for PW = 1..8
N_SGPR_LO, N_SGPR_HI = 40, 56
S_CUT = 56                    # f16p3s / mixs: 56..59 the second exit's masks and flag (N_SGPR_HI = 60 there)

SLOT = 32768
NSLOT = 4
LDS_AUX = NSLOT * SLOT
AUX_LAYER = 1280              # per layer: 272 f32 bias | at byte 1152: 4 lane quarters x (swl, sw, 0, 0)
AUX_SCALES = 1152
AUX_BYTES = 16384
LDS_BYTES = LDS_AUX + AUX_BYTES
ACT_EXP, RES_EXP = B.ACT_EXP, B.RES_EXP
# NERF_GEN_FMT=f16 (an environment variable: the tile / chunk / anchor tables below are built at import): the chain WITHOUT its
# correction terms -- one fp16 pass on the 256-wide sources (the embedding k-steps keep hi(E) and lo(E): XPASS below): no K=128 MFMA, no
# bf6 operand in the stream (1.27 MB per tile instead of 2.17, in 44 chunks instead of 80), no residuals and no 32-wide conversions in the epilogues.
# R2L_PREC_FP16X1 of the teacher: 1-3e-5 on rgb over whole frames (profiles/r04_teacher_x1.txt), `--precision auto`'s first rung.
# NERF_GEN_FMT=f16c3 / f16c4: the f16 chain with THREE / FOUR column tiles of 16 points per wave (192 / 256 points per workgroup tile):
# every weight fragment read from LDS feeds three / four MFMAs instead of two and the stream is fetched once per 192 / 256 points --
# what the bf6 chain has no registers for fits once the bf6 operand sets are gone: activation set P in VGPRs, set Q in
# AGPRs (an MFMA takes B from either file; Q's epilogues pay one v_accvgpr_write per packed register).  Four is the most that fits:
# 224 VGPRs + 16 outputs, 128 + 96 AGPRs.
# NERF_GEN_FMT=f16c4e (round 5): f16c4 as ONE asm statement that holds the tile loop -- rays and depths of the next tile are loaded
# by the stream itself (12 global loads into AGPRs at the start of a block), its embedding fragments are computed as FILLER
# instructions in the shadow of the MFMAs of L6 .. V (E in place once L5 has read it for the last time; V into the dead upper half
# of set P and from there into its AGPRs behind the V layer's last embedding k-step), and raw is stored by the stream: the HIP prologue
# that computed the embedding between two blocks while the matrix pipe idled (2.3 ms of a 33.4 ms frame, -DNERF_SKIP_EMBED) is gone.
# The arithmetic per point is that of nerf_tile_embed (csrc/nerf_kernels.hip), operation for operation: bitwise-equal results.
FMT = os.environ.get('NERF_GEN_FMT', 'bf6')
assert FMT in ('bf6', 'f16', 'f16c3', 'f16c4', 'f16c4e', 'f16p3', 'f16p3a', 'f16p3s', 'mix', 'mixs'), FMT
# NERF_GEN_FMT=mix (round 6): the bf6 chain with its FIRST trunk layers in three fp16 passes -- layers L1 .. L<MIXK> as in f16p3 (hi / lo
# fragments of both operands, W x 2^k in the stream), everything behind them with bf6 terms.  For the FINE pass of trained teachers: the
# early layers' error is what the sharp tail of such a network amplifies (tools/teacher_mixed_study.py on whole frames of the trained-like
# teacher, fine pass at fixed sample positions, L_inf from three passes everywhere: bf6 terms in every layer 1.2-1.6e-4, L1 in three
# passes 3.5-8.8e-5, L1 + L2 2.3-3.2e-5; profiles/r06_teacher_mixed_study.txt); the coarse pass, which steers sample_pdf, stays in f16p3.
# Per layer: Layer.p3.  L0 (embedding k-steps only: three passes in every format) hands L1 hi + lo fp16 sets, L<MIXK> hands the first
# bf6 layer hi + bf6 sets (its accumulators carry the 2^k of its stream: one v_mul more per value).  Registers: the bf6 map; the lo(a)
# sets of the three-pass layers live in AGPRs the bf6 chain does not use (lset).
# NERF_GEN_FMT=f16p3a (round 6): f16p3 WITHOUT the view branch -- trunk, then the alpha row of the feature | alpha layer alone, no views /
# rgb layers: raw = (0, 0, 0, sigma).  For the COARSE pass of a render whose caller does not ask for rgb0: sample_pdf and every map of the
# fine pass depend on the coarse network through its densities only (main.py:716-733), and the reference's call sites drop rgb0
# (main.py:277-282, utils/create_data.py:824-831: `rgb, disp, acc, _ = render(...)`).  17 % of the network's MACs; the alpha row gets its
# own weight scale 2^k.  A second, all-zero row tile keeps the chunk count a multiple of the ring's four slots (68).
ALPHA = FMT == 'f16p3a'
if ALPHA:
    FMT = 'f16p3'
# NERF_GEN_FMT=f16p3s / mixs (round 6): the chain with a SECOND EXIT behind the density.  A sample whose raw density is <= 0 has alpha = 0 and
# weight 0 exactly (main.py:600-606): its colour cannot reach rgb_map.  The feature | alpha layer is split -- layer A: the alpha row alone,
# FIRST; layer F: the 256 feature rows -- and behind A's epilogue the four waves of the workgroup agree (an OR through an LDS word, one
# barrier) whether ANY of the tile's 128 points has a positive density.  If none: the feature rows, the views layer and the rgb layer are dead
# work -- the block waits for the LDS-DMA pieces in flight, re-primes the ring with the next tile's first three chunks and leaves with
# raw = (0, 0, 0, sigma).  rgb / disp / acc / depth are bit for bit what the full chain gives (0 x sigmoid(c) = 0 either way); `raw` shows
# zeros for the colours of such tiles.  For renders whose caller drops the extras (nerf_set_skip_rgb0) and adds no density noise.
SKIPV = FMT in ('f16p3s', 'mixs')
if SKIPV:
    FMT = FMT[:-1]
    N_SGPR_HI = 60
MIX = FMT == 'mix'
MIXK = int(os.environ.get('NERF_GEN_MIX_K', '2')) if MIX else 0
assert 0 <= MIXK <= 7
X1 = FMT not in ('bf6', 'mix')
EMB = FMT == 'f16c4e'
# NERF_GEN_FMT=f16p3 (round 5): fp16x3's arithmetic on the generated chain -- per k-step three fp16 MFMAs on one accumulate chain,
# hi(W) hi(a) + hi(W) lo(a) + lo(W) hi(a), lo = the fp16 rounding residual -- for teachers that need fp32-grade arithmetic: every
# TRAINED one (profiles/r05_trained_like.txt: sharp densities amplify a single pass's 1e-5 to 3e-3, and the fine samples follow the
# coarse weights discontinuously).  Two column tiles per wave; lo(W) is a second set of streamed fragments (84 chunks of <= 32 KiB),
# lo(a) a second pair of activation sets in AGPRs; the stream holds W x 2^k (k per layer: max|w| 2^k in [2^12, 2^13)) so that lo(W) is
# a normal fp16 number, the epilogue takes the factor out (one v_fma_mix per value does it together with the conversion to fp16).
P3 = FMT == 'f16p3'
NC = {'f16c3': 3, 'f16c4': 4, 'f16c4e': 4}.get(FMT, 2)          # column tiles (16 points each) per wave
SUFFIX = {'bf6': '', 'f16': 'x', 'f16c3': 'x3', 'f16c4': 'x4', 'f16c4e': 'x4e', 'f16p3': 'p3', 'mix': 'm'}[FMT] + ('a' if ALPHA else '') + ('s' if SKIPV else '')     # nerf_mlpx_asm.inc ...
# passes of an embedding k-step: hi(W) hi(E), hi(W) lo(E), lo(W) hi(E).  The fp16-only chains drop the third (their 256-wide layers
# carry no lo(W) term either; measured over whole frames, three seed pairs x three poses: rgb 6.5e-6 .. 1.9e-5 from fp16x3 with two
# passes against 6.4e-6 .. 1.6e-5 with three, -4.4 % time; ONE pass -- no lo(E), i.e. fp16-rounded coordinates -- reads 1.3 .. 2.8e-5
# for another -4 % and is not taken).  The stream keeps the lo(W) fragments (unread: 72 of 1,298 KiB).  NERF_GEN_XPASS overrides.
XPASS = int(os.environ.get('NERF_GEN_XPASS', '2' if (X1 and not P3) else '3'))
MPASS = 3 if P3 else 1          # MFMAs per main k-step and column tile (per layer: Layer.mpass)
if NC > 2:
    V_SET = {'P': 0}              # fp16 B operands of set P (VGPR): + c*32 + s*4
    A_SETH = {'Q': 0}             # ... of set Q (AGPR)
    V_ACC = 32 * NC               # + p*4*NC + c*4
    V_BIAS = V_ACC + 8 * NC
    V_HI = V_BIAS + 8
    V_TMP = V_HI + 16             # + c*6
    V_A6 = V_LO = None
    _m = V_TMP + 6 * NC
    V_L0, V_L1, V_L8A, V_L8B, V_AUX, V_LANE = _m, _m + 1, _m + 2, _m + 3, _m + 4, _m + 5
    V_SBA = V_SBL = V_CVA = V_CVL = V_SC = None        # scales and divisors of the bf6 terms: not in this chain
    V_LOFF = _m + 6
    N_VGPR_CLOBBER = (_m + 7 + 3) // 4 * 4
    A_E = 32 * NC
    N_AGPR_CLOBBER = 32 * NC
else:
    A_SETH = {}
if P3:
    A_LOSET = {'P': 0, 'Q': 64}   # lo(a) B operands of the two activation sets (AGPR): + c*32 + s*4
    A_E = 128
    N_AGPR_CLOBBER = 128
    N_FRAG_BUF = 6                # rotating weight-fragment buffers from V_HI (hi and lo fragments alike): into V_A6's registers, unused here
else:
    N_FRAG_BUF = 4
# mix: AGPRs above the bf6 sets (0..95) that the block owns besides them: the last 16 registers of lo(Q) and of lo(P) (lset).  The
# compiler keeps the 48 input registers and what it carries across the block (the next tile's rays) out of both.
A_EXTRA_CLOBBER = (224, 256) if MIX else None
if EMB:
    V_OUT = N_VGPR_CLOBBER        # 16: (rgb, sigma) of the four column tiles -- and the embedding's temporaries until the RGB epilogue
    V_SIG = V_OUT + 16            # 4: sigma of the column tiles (FA's last row tile), moved into V_OUT at the tail
    V_G = V_SIG + 4               # 8: scratch of the address arithmetic (loads of the next tile's rays, stores of raw)
    V_PL = V_G + 8                # wave * 64 + (lane & 15): the lane's point inside a column tile of the workgroup tile
    assert V_PL < 256
    N_VGPR_CLOBBER = 256
    A_RAW = A_E + 24 * NC         # 32: per column tile c  o[3] | z at + 4 c,  d[3] at + 16 + 4 c  (even-aligned dwordx3 destinations)
    N_AGPR_CLOBBER = 256
    assert A_RAW + 32 == 256
    S_PO, S_PD, S_PZ, S_PRAW = 56, 58, 60, 62          # pointers: rays_o, rays_d, z, raw
    S_NPTS, S_LAST, S_S, S_ZS, S_MAGIC, S_SH1, S_SH2 = 64, 65, 66, 67, 68, 69, 70
    S_TILE, S_NEXT, S_GRID, S_NTILES, S_TB = 71, 72, 73, 74, 75
    S_C25, S_HI, S_LO, S_A16 = 76, 77, 78, 79          # 0.25, fl32(1 / 2 pi), 1 / 2 pi - fl32(1 / 2 pi)  (csrc/r2l_device.h), act_scale
    S_MQ1, S_MQ2, S_MQ3, S_MQ0, S_MQ1E, S_ML16, S_TM = 80, 82, 84, 86, 88, 90, 92      # lane masks: q & 1, q & 2, q == 3, q == 0, q == 1, lane < 16, scratch
    N_SGPR_HI = 96


class Layer:
    def __init__(self, name, src, dst, ks, extra, rt, epi, rt_per_chunk, fan_out):
        self.name, self.src, self.dst, self.ks, self.extra, self.rt, self.epi = name, src, dst, ks, extra, rt, epi
        self.rt_per_chunk = rt_per_chunk
        self.fan_out = fan_out           # real output rows
        self.p3 = P3                     # main k-steps as three fp16 passes (mix: set per layer by chain())
        self.nj = 0 if X1 else ks // 2   # K=128 MFMAs per row tile and column tile (2 terms x ks/4)
        self.nx = len(extra)

    def set_p3(self, on):
        self.p3 = bool(on)
        self.nj = 0 if (X1 or self.p3) else self.ks // 2

    @property
    def mpass(self):
        return 3 if self.p3 else 1

    def j_order(self):
        """(term, t) of the K=128 MFMAs in issue order: term 0 = (w - hi) x bf6(a), 1 = w x bf6(a - hi)"""
        if X1 or self.p3:
            return []
        return [(0, 0), (1, 0), (0, 1), (1, 1)] if self.ks == 8 else [(0, 0), (1, 0)]

    def chunk_pieces(self):
        r = self.rt_per_chunk
        if self.p3:
            return r * self.ks * 2 + r * self.nx * 2
        return r * self.ks + r * self.nj + (r * self.nj + 1) // 2 + r * self.nx * 2


def chain():
    E2 = [('E', 0), ('E', 1)]
    # row tiles per chunk: a chunk must fit a 32 KiB ring slot; without the bf6 operands twice as many row tiles do (44 chunks and
    # rendezvous per tile instead of 80)
    r_std, r_l5, r_v = (2, 1, 1) if P3 else ((4, 2, 2) if X1 else (2, 1, 2))
    L = [Layer('L0', None, 'Q', 0, E2, 16, 'relu', 8, 256)]
    sets = ['Q', 'P']
    for i in range(1, 5):
        L.append(Layer('L%d' % i, sets[(i - 1) % 2], sets[i % 2], 8, [], 16, 'relu', r_std, 256))
    L.append(Layer('L5', 'Q', 'P', 8, E2, 16, 'relu', r_l5, 256))
    L.append(Layer('L6', 'P', 'Q', 8, [], 16, 'relu', r_std, 256))
    L.append(Layer('L7', 'Q', 'P', 8, [], 16, 'relu', r_std, 256))
    if ALPHA:       # the alpha row alone (row tile 0, row 0) + an all-zero row tile (chunk count = 0 mod 4); no view branch
        L.append(Layer('FA', 'P', None, 8, [], 2, 'alpha', 1, 1))
    elif SKIPV:     # the alpha row first (one row tile: one chunk), then the cut, then the feature rows (8 chunks: 9 as the unsplit layer)
        L.append(Layer('A', 'P', None, 8, [], 1, 'alpha', 1, 1))
        L.append(Layer('F', 'P', 'Q', 8, [], 16, 'feat', r_std, 256))
        L.append(Layer('V', 'Q', 'P', 8, [('V', 0)], 8, 'relu', r_v, 128))
        L.append(Layer('RGB', 'P', None, 4, [], 1, 'rgb', 1, 3))
    else:
        L.append(Layer('FA', 'P', 'Q', 8, [], 17, 'feat', r_std, 257))
        L.append(Layer('V', 'Q', 'P', 8, [('V', 0)], 8, 'relu', r_v, 128))
        L.append(Layer('RGB', 'P', None, 4, [], 1, 'rgb', 1, 3))
    for li, l in enumerate(L):
        l.li = li
        if MIX:
            l.set_p3(1 <= li <= MIXK)
            if l.p3:
                l.rt_per_chunk = {'L5': 1}.get(l.name, 2)
    for li, l in enumerate(L):
        # exp_group: layers whose bf6 terms share one weight exponent (A and F: that of the unsplit feature | alpha layer, so that the split
        # chain computes bit for bit what the unsplit one does)
        l.exp_group = ('A', 'F') if l.name in ('A', 'F') else (l.name,)
    for li, l in enumerate(L):
        # lo_out: the epilogue hands its consumer hi + lo fp16 sets (the consumer runs three passes); scaled: its accumulators carry the
        # 2^k of a three-pass stream; uses_inv: its epilogue multiplies by the f32 in the layer's scale registers (1 / 2^k; L0 of mix: 1.0)
        l.lo_out = li + 1 < len(L) and L[li + 1].p3
        l.scaled = l.p3
        l.uses_inv = l.p3 or (MIX and l.lo_out)
    return L


CHAIN = chain()


class Tile:
    def __init__(self, idx, li, u):
        self.idx, self.li, self.u, self.layer = idx, li, u, CHAIN[li]


def _tiles_and_chunks():
    tiles, chunks = [], []
    for li, L in enumerate(CHAIN):
        u = 0
        while u < L.rt:
            n = min(L.rt_per_chunk, L.rt - u)
            ids = []
            for k in range(n):
                ids.append(len(tiles))
                tiles.append(Tile(len(tiles), li, u + k))
            pieces = L.chunk_pieces()
            chunks.append(dict(tiles=ids, pieces=pieces, pw=(pieces + 3) // 4, li=li))
            u += n
    return tiles, chunks


TILES, CHUNKS = _tiles_and_chunks()
NT = len(TILES)
NCH = len(CHUNKS)
assert NCH % NSLOT == 0, NCH
assert max(c['pw'] for c in CHUNKS) * 4096 <= SLOT
CHUNK_OFF = [0]
for _c in CHUNKS:
    CHUNK_OFF.append(CHUNK_OFF[-1] + _c['pw'] * 4096)
STREAM_BYTES = CHUNK_OFF[-1]
TILE_CHUNK = {}
for _ci, _c in enumerate(CHUNKS):
    for _k, _t in enumerate(_c['tiles']):
        TILE_CHUNK[_t] = (_ci, _k)
TILE_OF = {(t.li, t.u): t.idx for t in TILES}


def piece_of(L, k, what, idx):
    """(1 KiB piece, byte offset inside it) within its chunk of an operand of the k-th row tile of the chunk.  what: 'hi'
    (fp16 fragment of main k-step idx), 'b6' / 'b6b' (first 16 / last 8 bytes per lane of bf6 operand idx), 'xh' / 'xl'
    (hi / lo fragment of embedding k-step idx)"""
    r = L.rt_per_chunk
    if L.p3:    # per row tile: its ks hi fragments, then its ks lo fragments; behind all row tiles the embedding k-steps (hi, lo)
        if what in ('hi', 'lo'):
            return k * L.ks * 2 + (L.ks if what == 'lo' else 0) + idx, 0
        return r * L.ks * 2 + (k * L.nx + idx) * 2 + (1 if what == 'xl' else 0), 0
    if what == 'hi':
        return k * L.ks + idx, 0
    base = r * L.ks
    if what == 'b6':
        return base + k * L.nj + idx, 0
    base += r * L.nj
    if what == 'b6b':
        o = (k * L.nj + idx) * 512
        return base + o // 1024, o % 1024
    base += (r * L.nj + 1) // 2
    if what == 'xh':
        return base + (k * L.nx + idx) * 2, 0
    if what == 'xl':
        return base + (k * L.nx + idx) * 2 + 1, 0
    raise ValueError(what)


# ---------------------------------------------------------------------------------------------
# embedding layouts (nerf_common.h restated) and the host packer restated
# ---------------------------------------------------------------------------------------------
def pts_col(e, q, j):
    if e == 0:
        return 3 + j * 6 + (3 if (q & 1) else 0) + (q >> 1)
    if q < 2:
        return 3 + j * 6 + (3 if (q & 1) else 0) + 2
    if j < 6:
        return 3 + (8 + (j & 1)) * 6 + (3 if q == 3 else 0) + (j >> 1)
    if q == 2:
        return j - 6
    return 2 if j == 6 else -1


def view_col(q, j):
    if q < 3:
        return 3 + (j & 3) * 6 + (3 if (j >> 2) else 0) + q
    return j if j < 3 else -1


def layer_matrices(t):
    """per CHAIN layer (W_main [rows, K] or None, W_emb [rows, 63 | 27] or None, bias) from the 24 state_dict tensors
    (model/nerf_raybased.py:357-375 order), float32 numpy"""
    t = [np.asarray(x, dtype=np.float32) for x in t]
    W = lambda i, r, c: t[i].reshape(r, c)
    out = [(None, W(0, 256, 63), t[1])]
    for i in range(1, 5):
        out.append((W(2 * i, 256, 256), None, t[2 * i + 1]))
    w5 = W(10, 256, 319)                                                   # cat([input_pts, h]) (:385)
    out.append((np.ascontiguousarray(w5[:, 63:]), np.ascontiguousarray(w5[:, :63]), t[11]))
    for i in (6, 7):
        out.append((W(2 * i, 256, 256), None, t[2 * i + 1]))
    if ALPHA:
        out.append((W(20, 1, 256), None, t[21].reshape(-1)))
        return out
    if SKIPV:
        out.append((W(20, 1, 256), None, t[21].reshape(-1)))
        out.append((W(18, 256, 256), None, t[19].reshape(-1)))
    else:
        out.append((np.concatenate([W(18, 256, 256), W(20, 1, 256)], 0), None, np.concatenate([t[19].reshape(-1), t[21].reshape(-1)])))
    wv = W(16, 128, 283)                                                   # cat([feature, input_views]) (:390)
    out.append((np.ascontiguousarray(wv[:, :256]), np.ascontiguousarray(wv[:, 256:]), t[17]))
    out.append((W(22, 3, 128), None, t[23]))
    return out


def pack_teacher(tensors, act_scale=16.0):
    """(stream bytes of one tile [STREAM_BYTES], aux bytes [AUX_BYTES]): Python restatement of nerf_capi.hip
    pack_teacher_bf6"""
    mats = layer_matrices(tensors)
    img = np.zeros(STREAM_BYTES, dtype=np.uint8)
    aux = np.zeros(AUX_BYTES // 4, dtype=np.uint32)
    lanes = np.arange(64)
    q, r = lanes >> 4, lanes & 15
    for li, L in enumerate(CHAIN):
        Wm, We, bias = mats[li]
        a0 = li * AUX_LAYER // 4
        aux[a0:a0 + len(bias)] = (bias.astype(np.float64) * act_scale).astype(np.float32).view(np.uint32)
        el = ew = 0
        if L.uses_inv and not L.p3:     # mix, L0: its epilogue is the three-pass one (hi + lo sets for L1) with factor 1
            for qq in range(4):
                aux[a0 + AUX_SCALES // 4 + 4 * qq] = np.array([1.0], dtype=np.float32).view(np.uint32)[0]
        if L.p3:    # W x 2^k with max|w| 2^k in [2^12, 2^13) over everything that accumulates into this layer's rows (r2l_pow2_scale)
            grp = [m for j, l2 in enumerate(CHAIN) if l2.name in L.exp_group for m in mats[j][:2] if m is not None]    # (A, F: the unsplit layer's)
            mx = max(float(np.abs(m).max()) for m in grp)
            sw = np.float32(2.0 ** (12 - int(np.floor(np.log2(mx)))) if mx > 0 and np.isfinite(mx) else 1.0)
            aux[a0:a0 + len(bias)] = (bias.astype(np.float64) * act_scale * float(sw)).astype(np.float32).view(np.uint32)
            for qq in range(4):
                aux[a0 + AUX_SCALES // 4 + 4 * qq] = np.array([1.0 / sw], dtype=np.float32).view(np.uint32)[0]
            Wm = None if Wm is None else (Wm * sw).astype(np.float32)
            We = None if We is None else (We * sw).astype(np.float32)
        if Wm is not None:
            hi = Wm.astype(np.float16)
        if Wm is not None and not X1 and not L.p3:       # the E8M0 scale bytes of the bf6 terms
            grp = [mats[j][0] for j, l2 in enumerate(CHAIN) if l2.name in L.exp_group]
            el, ew = weight_exps(max(layer_exponent(g) for g in grp))
            for qq in range(4):
                aux[a0 + AUX_SCALES // 4 + 4 * qq] = 0x01010101 * (127 + el)
                aux[a0 + AUX_SCALES // 4 + 4 * qq + 1] = 0x01010101 * (127 + ew)
        for u in range(L.rt):
            ci, k = TILE_CHUNK[TILE_OF[(li, u)]]
            base = CHUNK_OFF[ci]
            rows = 16 * u + r
            ok = rows < L.fan_out
            rws = np.where(ok, rows, 0)

            def put(what, idx, data, nbytes=1024):
                p, o = piece_of(L, k, what, idx)
                a = base + p * 1024 + o
                img[a:a + nbytes] = np.ascontiguousarray(data).view(np.uint8).reshape(-1)

            for s in range(L.ks):
                frag = np.zeros((64, 8), dtype=np.float16)
                for j in range(8):
                    frag[:, j] = np.where(ok, hi[rws, kappa(s, q, j)], 0)
                put('hi', s, frag)
                if L.p3:
                    frag = np.zeros((64, 8), dtype=np.float16)
                    for j in range(8):
                        kk = kappa(s, q, j)
                        frag[:, j] = np.where(ok, (Wm[rws, kk] - hi[rws, kk].astype(np.float32)).astype(np.float16), 0)
                    put('lo', s, frag)
            for j, (term, tt) in enumerate(L.j_order() if L.ks else []):
                codes = np.zeros((64, 32), dtype=np.uint8)
                for e in range(32):
                    kk = mix_feat(tt, q, e)
                    w = np.where(ok, Wm[rws, kk], 0).astype(np.float64)
                    h = np.where(ok, hi[rws, kk], 0).astype(np.float64)
                    codes[:, e] = f_to_bf6(np.ldexp(w - h, -el) if term == 0 else np.ldexp(w, -ew))
                words = pack6(codes)
                put('b6', j, words[:, :4])
                put('b6b', j, words[:, 4:], 512)
            for xi, (kind, e) in enumerate(L.extra):
                fh = np.zeros((64, 8), dtype=np.float16)
                fl = np.zeros((64, 8), dtype=np.float16)
                for j in range(8):
                    for qq in range(4):
                        col = pts_col(e, qq, j) if kind == 'E' else view_col(qq, j)
                        if col < 0:
                            continue
                        m = (q == qq) & ok
                        w = We[rws, col].astype(np.float32)
                        h = w.astype(np.float16)
                        fh[m, j] = h[m]
                        fl[m, j] = (w - h.astype(np.float32)).astype(np.float16)[m]
                put('xh', xi, fh)
                put('xl', xi, fl)
    return img, aux.view(np.uint8)


# ---------------------------------------------------------------------------------------------
# builders beyond body_gen's
# ---------------------------------------------------------------------------------------------
def E_reg(kind, e, c, lo):
    if kind == 'E':
        return A_E + (8 * NC if lo else 0) + (e * NC + c) * 4
    return A_E + 16 * NC + (4 * NC if lo else 0) + c * 4


def E_name(kind, e, c, lo):
    if kind == 'E':
        return '%%[e%s%d%d]' % ('l' if lo else 'h', e, c)
    return '%%[v%s%d]' % ('l' if lo else 'h', c)


INPUT_NAMES = ([('e%s%d%d' % (hl, e, c), E_reg('E', e, c, hl == 'l')) for hl in 'hl' for e in range(2) for c in range(NC)] +
               [('v%s%d' % (hl, c), E_reg('V', 0, c, hl == 'l')) for hl in 'hl' for c in range(NC)])


def mfma16(d, a, bfile, b, c, tag='', btext=None):
    """v[d:d+3] = A(v[a:a+3]) x B([bfile] b:b+3) + v[c:c+3]; btext: asm operand placeholder of an AGPR input"""
    rf = {'v': vreg, 'a': areg}
    text = 'v_mfma_f32_16x16x32_f16 %s, %s, %s, %s' % (vreg(d, 4), vreg(a, 4), btext or rf[bfile](b, 4), vreg(c, 4))

    def emu(st):
        lanes = np.arange(64)
        Ah = B._halves(st.V[a:a + 4])
        Bh = B._halves(st.regs(bfile)[b:b + 4])
        Am = np.zeros((16, 32))
        Bm = np.zeros((32, 16))
        for j in range(8):
            Am[lanes & 15, 8 * (lanes >> 4) + j] = Ah[:, j]
            Bm[8 * (lanes >> 4) + j, lanes & 15] = Bh[:, j]
        D = Am @ Bm
        C = st.V[c:c + 4].view(np.float32).astype(np.float64)
        out = np.zeros((4, 64), dtype=np.float32)
        for i in range(4):
            out[i] = (C[i] + D[4 * (lanes >> 4) + i, lanes & 15]).astype(np.float32)
        st.V[d:d + 4] = out.view(np.uint32)

    rb = vr(b, 4) if bfile == 'v' else ar(b, 4)
    return Ins(text, 'mfma16', rd=vr(a, 4) + rb + vr(c, 4), wr=vr(d, 4), emu=emu, tag=tag)


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


def mfma6(d, a, b_agpr, scale_a, scale_b, tag=''):
    return B.mfma6('v', d, a, b_agpr, scale_a, scale_b, tag)


def v_mixs(dst, high, a, scale_v, hsrc=None, hhalf=0):
    """half `high` of dst = f16(a * v[scale_v] - h), h = 0 or the f16 half `hhalf` of v[hsrc] (v_fma_mixlo/hi_f16: f32 a, f32 scale,
    f16 third source): the epilogue of f16p3 -- conversion to fp16, removal of the layer's weight scale and the residual in one
    instruction per half"""
    op = 'v_fma_mixhi_f16' if high else 'v_fma_mixlo_f16'
    if hsrc is None:
        text = '%s %s, %s, %s, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]' % (op, vreg(dst), vreg(a), vreg(scale_v))
    else:
        text = '%s %s, %s, %s, -%s op_sel:[0,0,%d] op_sel_hi:[0,0,1]' % (op, vreg(dst), vreg(a), vreg(scale_v), vreg(hsrc), hhalf)

    def emu(st):
        h = np.zeros(64) if hsrc is None else ((st.V[hsrc] >> (16 * hhalf)) & 0xffff).astype(np.uint16).view(np.float16).astype(np.float64)
        r = (st.f32('v', a).astype(np.float64) * st.f32('v', scale_v).astype(np.float64) - h).astype(np.float32).astype(np.float16)
        r = r.view(np.uint16).astype(np.uint32)
        st.V[dst] = (st.V[dst] & 0x0000ffff) | (r << 16) if high else (st.V[dst] & 0xffff0000) | r
    return valu(text, vr(a) + vr(scale_v) + (vr(hsrc) if hsrc is not None else []), vr(dst), emu, partial=True)


def v_mov_out(k, src, inv=None):
    """output operand k <- v[src] (the emulator keeps outputs in st.out).  f16c4e: the outputs are the stream's own registers --
    rgb straight into V_OUT (RGB's exposed epilogue: the embedding's temporaries are free by then), sigma, which FA's last row
    tile delivers while those temporaries are in use, into V_SIG"""
    if EMB:
        dst = V_SIG + k // 4 if k % 4 == 3 else V_OUT + k
        return B.v_mov_b32(dst, ('v', src))
    if inv is not None:       # f16p3: the accumulator carries the layer's weight scale 2^k; v[inv] = 2^-k

        def emu_s(st):
            st.out[k] = (st.f32('v', src) * st.f32('v', inv)).astype(np.float32).view(np.uint32)
        return valu('v_mul_f32 %%[o%d], %s, %s' % (k, vreg(inv), vreg(src)), vr(src) + vr(inv), [], emu_s)

    def emu(st):
        st.out[k] = st.V[src].copy()
    return valu('v_mov_b32 %%[o%d], %s' % (k, vreg(src)), vr(src), [], emu)


def v_zero_out(k):
    """output operand k <- 0.0"""
    def emu(st):
        st.out[k] = np.zeros(64, dtype=np.uint32)
    return valu('v_mov_b32 %%[o%d], 0' % k, [], [], emu)


def dma_piece(i, pw, tag=''):
    """piece i of this wave's pw KiB of a chunk: global_load_lds_dwordx4 v_off, s[S_G:S_G+1] offset:imm with LDS
    destination M0 + imm + lane*16; pieces 4.. use the +4096 offset register and M0 + 4096"""
    voffr = V_L0 if i < 4 else V_LOFF
    imm = 1024 * (i & 3)
    text = 'global_load_lds_dwordx4 %s, %s offset:%d' % (vreg(voffr), sreg(S_G, 2), imm)

    def emu(st):
        copies = []
        g = st.S[S_G]
        for w in range(4):
            dw = (w - st.wave) * pw * 1024
            for l in range(64):
                src = g + int(st.V[voffr][l]) + dw + imm
                dst = st.m0 + dw + imm + l * 16
                assert 0 <= dst and dst + 16 <= LDS_AUX, dst
                assert 0 <= src and src + 16 <= len(st.img), (src, len(st.img))
                copies.append((dst, st.img[src:src + 16].copy()))
                st.lds_pending[dst:dst + 16] = True
        st.pend_dma.append(copies)
        st.vmq.append(('d', None))
    return Ins(text, 'dma', rd=vr(voffr), emu=emu, cost=8, tag=tag)


class NState(B.State):
    def __init__(self, wave, img, aux):
        B.State.__init__(self, wave, img, np.zeros((1, 1024), dtype=np.uint32), 0, LDS_BYTES)
        self.lds[LDS_AUX:LDS_AUX + len(aux)] = aux
        self.out = {}
        self.vmq = []                       # f16c4e: every vector memory operation in issue order ('d' LDS-DMA piece, 'v' register load, 'st' store)
        self.gmem = {}                      # ... and the global buffers its loads and stores see: name -> flat numpy array
        self.vcc = np.zeros(64, bool)
        self.exec_ = np.ones(64, bool)


# ---------------------------------------------------------------------------------------------
# anchors: the MFMAs in program order
# ---------------------------------------------------------------------------------------------
def tile_anchors(L):
    """MFMAs of one row tile as (kind, k, c, p): ('x', xi, c, pass) of the embedding k-steps first (pass 0 Wh x Eh,
    1 Wh x El, 2 Wl x Eh), then ('m16', s, c, 0) and, behind the last nj main k-steps, ('m6', j, c, 0)"""
    out = []
    for xi in range(L.nx):
        for p in range(XPASS):
            for c in range(NC):
                out.append(('x', xi, c, p))
    for s in range(L.ks):
        for p in range(L.mpass):        # three passes: hi(W) hi(a), hi(W) lo(a), lo(W) hi(a)
            for c in range(NC):
                out.append(('m16', s, c, p))
        j = s - (L.ks - L.nj)
        if j >= 0:
            out.append(('m6', j, 0, 0))
            out.append(('m6', j, 1, 0))
    return out


ANCH = [tile_anchors(t.layer) for t in TILES]
ABASE = [0]
for _a in ANCH:
    ABASE.append(ABASE[-1] + len(_a))
N_ANCH = ABASE[-1]


def aidx(T, kind, k, c, p=0):
    return ABASE[T] + ANCH[T].index((kind, k, c, p))


def afirst(T):
    return ABASE[T] if T < NT else N_ANCH


def operand_key(T, kind, k, p):
    if kind == 'x':
        return ('xl' if p == 2 else 'xh', T, k)
    if kind == 'm16':
        return ('lo' if p == 2 else 'hi', T, k)
    return ('a6', T, k)


def hfile(name):
    """register file of activation set `name` (f16c3: Q lives in AGPRs)"""
    return 'a' if name in A_SETH else 'v'


def hset(name, s, c):
    return (A_SETH[name] if name in A_SETH else V_SET[name]) + c * 32 + s * 4


def lset(name, s, c):
    """lo(a) B operand of activation set `name` (f16p3: AGPR)"""
    if MIX:
        # during L1 lo(Q) (read) and lo(P) (written) are both live and no bf6 set is: each takes one bf6 set's 48 registers (lo(Q) those of
        # bf6(Q), which L2's epilogues write first, when L1 is over; lo(P) those of bf6(P), written first by L3's epilogues) and 16 above
        # the inputs
        i = c * 8 + s
        if name == 'Q':
            return 48 + i * 4 if i < 12 else 224 + (i - 12) * 4
        return i * 4 if i < 12 else 240 + (i - 12) * 4
    return A_LOSET[name] + c * 32 + s * 4


def b6(name, term, t, c):
    return A_SET[name] + term * 24 + (t * 2 + c) * 6


def ACC(p, c):
    return V_ACC + p * 4 * NC + c * 4


def LO(c):
    return V_LO + c * 16


def TMP(c):
    return V_TMP + c * (10 if NC == 2 else 6)


def epilogue_ops(T, c):
    """epilogue of tile T for column tile c: [(Ins, consumer)], consumer None | ('hi', s) | ('b6', term, t)"""
    t = TILES[T]
    L, u = t.layer, t.u
    acc = ACC(T & 1, c)
    tb = TMP(c)
    cv = tb + 4
    ops = []
    inv = V_SC + 2 * (t.li & 1) if L.uses_inv else None
    if L.epi == 'alpha':     # f16p3a: sigma from row 0 of row tile 0, zeros for the colours nobody computed; the padding row tile: nothing
        if u != 0:
            return []
        if SKIPV:            # the colours come from the rgb layer, or as zeros from the second exit
            return [(v_mov_out(c * 4 + 3, acc, inv), None)]
        return [(v_mov_out(c * 4 + 3, acc, inv), None)] + [(v_zero_out(c * 4 + k), None) for k in range(3)]
    if L.epi == 'rgb':
        return [(v_mov_out(c * 4 + k, acc + k, inv), None) for k in range(3)]
    if L.epi == 'feat' and u == 16:
        return [(v_mov_out(c * 4 + 3, acc, inv), None)]
    if L.epi == 'relu':
        tv = [tb + i for i in range(4)]
        for i in range(4):
            ops.append((v_max0(tv[i], acc + i), None))
        if L.scaled and not L.lo_out:       # mix, the last three-pass layer: the bf6-style epilogue below on values that carry 2^k
            for i in range(4):
                ops.append((v_mul(tv[i], tv[i], inv), None))
    else:
        assert not (L.scaled and not L.lo_out)
        tv = [acc + i for i in range(4)]
    h01 = hset(L.dst, u >> 1, c) + 2 * (u & 1)
    h23 = h01 + 1
    if L.lo_out if MIX else P3:
        # per value one v_fma_mix for the hi half -- f16(t 2^-k) -- and one for the lo half -- f16(t 2^-k - hi) --; the lo pairs go
        # through two temporaries into the AGPR set.  No two consecutive instructions touch a register one of them half-writes.
        l01, l23 = tb + 4, tb + 5
        lo01 = lset(L.dst, u >> 1, c) + 2 * (u & 1)
        for dst, high, src, hs, hh, cons in ((h01, 0, tv[0], None, 0, None), (h23, 0, tv[2], None, 0, None),
                                              (h01, 1, tv[1], None, 0, ('hi', u >> 1)), (h23, 1, tv[3], None, 0, ('hi', u >> 1)),
                                              (l01, 0, tv[0], h01, 0, None), (l23, 0, tv[2], h23, 0, None),
                                              (l01, 1, tv[1], h01, 1, None), (l23, 1, tv[3], h23, 1, None)):
            ops.append((v_mixs(dst, high, src, inv, hs, hh), cons))
        ops.append((v_accw(lo01, l01), ('lo', u >> 1)))
        ops.append((v_accw(lo01 + 1, l23), ('lo', u >> 1)))
        return ops
    if hfile(L.dst) == 'a':       # the packed pairs go through two temporaries into the AGPR set
        p01, p23 = tb + 4, tb + 5
        ops.append((v_cvt_pk_f16(p01, tv[0], tv[1]), None))
        ops.append((v_cvt_pk_f16(p23, tv[2], tv[3]), None))
        ops.append((v_accw(h01, p01), ('hi', u >> 1)))
        ops.append((v_accw(h23, p23), ('hi', u >> 1)))
        return ops
    ops.append((v_cvt_pk_f16(h01, tv[0], tv[1]), ('hi', u >> 1)))
    ops.append((v_cvt_pk_f16(h23, tv[2], tv[3]), ('hi', u >> 1)))
    if X1:
        return ops
    lo = LO(c) + 2 * (u & 7)
    # half-register writes: low halves first, then the high halves (never two writers of one register back to back)
    ops.append((v_resid16(lo, 0, h01, 0, tv[0]), None))
    ops.append((v_resid16(lo + 1, 0, h23, 0, tv[2]), None))
    ops.append((v_resid16(lo, 1, h01, 1, tv[1]), None))
    ops.append((v_resid16(lo + 1, 1, h23, 1, tv[3]), None))
    if (u & 7) == 7:
        tt = u >> 3
        ops.append((v_cvt_pk32_bf6(cv, hset(L.dst, 4 * tt, c), V_CVA), None))
        for i in range(6):
            ops.append((v_accw(b6(L.dst, 0, tt, c) + i, cv + i), ('b6', 0, tt)))
        ops.append((v_cvt_pk32_bf6(cv, LO(c), V_CVL), None))
        for i in range(6):
            ops.append((v_accw(b6(L.dst, 1, tt, c) + i, cv + i), ('b6', 1, tt)))
    return ops


# ---------------------------------------------------------------------------------------------
# f16c4e: the embedding, the ray loads and the raw stores as part of the stream
# ---------------------------------------------------------------------------------------------
INV2PI_HI = np.float32(0.15915494)            # csrc/r2l_device.h R2L_INV2PI_HI / _LO
INV2PI_LO = np.float32(6.4206382e-09)
ACT = 16.0                                    # act_scale of the chain (nerf_capi.hip: c->act_scale)


def RAW_O(c):
    return A_RAW + 4 * c


def RAW_Z(c):
    return A_RAW + 4 * c + 3


def RAW_D(c):
    return A_RAW + 16 + 4 * c


def _vv(n):
    return ('v', n)


def _mask(st, s):
    return st.S[s]


def v_cndmask(dst, a, b, smask):
    """dst = mask ? b : a per lane; a, b: VGPR numbers or 0; mask: SGPR pair (VOP3)"""
    ta = vreg(a) if a is not None else '0'
    tb = vreg(b) if b is not None else '0'

    def emu(st):
        va = st.V[a] if a is not None else np.zeros(64, np.uint32)
        vb = st.V[b] if b is not None else np.zeros(64, np.uint32)
        st.V[dst] = np.where(_mask(st, smask), vb, va).astype(np.uint32)
    rd = ([('v', a)] if a is not None else []) + ([('v', b)] if b is not None else [])
    return valu('v_cndmask_b32_e64 %s, %s, %s, %s' % (vreg(dst), ta, tb, sreg(smask, 2)), rd, vr(dst), emu)


def v_sub_abs(dst, sconst, a):
    """dst = s[sconst] - |a|"""
    def emu(st):
        st.V[dst] = (np.float32(st.S[sconst]) - np.abs(st.f32('v', a))).astype(np.float32).view(np.uint32)
    return valu('v_sub_f32_e64 %s, %s, |%s|' % (vreg(dst), sreg(sconst), vreg(a)), vr(a), vr(dst), emu)


def v_fma_na(dst, a, b, c):
    """dst = fma(-a, b, c); a, b VGPRs, c a VGPR or the inline constant 1.0"""
    tc = vreg(c) if isinstance(c, int) else repr(float(c))

    def emu(st):
        cc = st.f32('v', c).astype(np.float64) if isinstance(c, int) else np.float64(c)
        st.V[dst] = (-(st.f32('v', a).astype(np.float64) * st.f32('v', b).astype(np.float64)) + cc).astype(np.float32).view(np.uint32)
    return valu('v_fma_f32 %s, -%s, %s, %s' % (vreg(dst), vreg(a), vreg(b), tc), vr(a) + vr(b) + (vr(c) if isinstance(c, int) else []), vr(dst), emu)


def v_fmac(dst, a, b):
    """dst = a * b + dst (one rounding)"""
    def emu(st):
        st.V[dst] = (st.f32('v', a).astype(np.float64) * st.f32('v', b).astype(np.float64) + st.f32('v', dst).astype(np.float64)
                     ).astype(np.float32).view(np.uint32)
    return valu('v_fmac_f32_e32 %s, %s, %s' % (vreg(dst), vreg(a), vreg(b)), vr(a) + vr(b) + vr(dst), vr(dst), emu)


def _trans(text, rd, wr, emu):
    ins = valu(text, rd, wr, emu)
    ins.kind = 'trans'
    ins.cost = 4
    return ins


def v_sin(dst, a):
    ins = B.v_sin_f32(dst, _vv(a))
    ins.cost = 4
    return ins


def v_sqrt(dst, a):
    def emu(st):
        st.V[dst] = np.sqrt(st.f32('v', a)).astype(np.float32).view(np.uint32)
    return _trans('v_sqrt_f32_e32 %s, %s' % (vreg(dst), vreg(a)), vr(a), vr(dst), emu)


def v_rcp(dst, a):
    def emu(st):
        st.V[dst] = (np.float32(1.0) / st.f32('v', a)).astype(np.float32).view(np.uint32)
    return _trans('v_rcp_f32_e32 %s, %s' % (vreg(dst), vreg(a)), vr(a), vr(dst), emu)


def v_addi(dst, imm, a):
    """integer dst = a + imm (the neighbours of a float in v_sqrt's correction step)"""
    def emu(st):
        st.V[dst] = (st.V[a].astype(np.int64) + imm).astype(np.uint32)
    return valu('v_add_u32_e32 %s, %d, %s' % (vreg(dst), imm, vreg(a)), vr(a), vr(dst), emu)


def v_cmp0(op, smask, a):
    """s[smask] = (0 op a) per lane, op in ge / lt"""
    fn = {'ge': lambda x: 0.0 >= x, 'lt': lambda x: 0.0 < x}[op]

    def emu(st):
        st.S[smask] = fn(st.f32('v', a))
    return valu('v_cmp_%s_f32_e64 %s, 0, %s' % (op, sreg(smask, 2), vreg(a)), vr(a), [], emu)


def v_div_scale(dst, sdst, s0, s1, s2):
    """v_div_scale_f32 dst, sdst, s0, s1, s2 (s1 denominator, s2 numerator; returns s0 scaled).  The emulator models the case the
    stream is used in -- no operand near the ends of the exponent range: nothing is scaled, the flag is clear"""
    def emu(st):
        x = st.f32('v', s0)
        ok = np.isfinite(x) & ((np.abs(x) > 2.0 ** -60) | (x == 0)) & (np.abs(x) < 2.0 ** 60)
        if not ok.all():
            st.errors.append('ins %d: v_div_scale_f32 operand outside the modelled range' % st.n_ins)
        st.V[dst] = st.V[s0]
        if sdst == 'vcc':
            st.vcc = np.zeros(64, bool)
        else:
            st.S[sdst] = np.zeros(64, bool)
    sd = 'vcc' if sdst == 'vcc' else sreg(sdst, 2)
    return valu('v_div_scale_f32 %s, %s, %s, %s, %s' % (vreg(dst), sd, vreg(s0), vreg(s1), vreg(s2)), vr(s0) + vr(s1) + vr(s2), vr(dst), emu)


def v_div_fmas(dst, a, b, c):
    def emu(st):
        assert not st.vcc.any()
        st.V[dst] = (st.f32('v', a).astype(np.float64) * st.f32('v', b).astype(np.float64) + st.f32('v', c).astype(np.float64)
                     ).astype(np.float32).view(np.uint32)
    return valu('v_div_fmas_f32 %s, %s, %s, %s' % (vreg(dst), vreg(a), vreg(b), vreg(c)), vr(a) + vr(b) + vr(c), vr(dst), emu)


def v_div_fixup(dst, q, den, num):
    def emu(st):
        st.V[dst] = st.V[q]          # finite, non-zero operands: the quotient passes through
    return valu('v_div_fixup_f32 %s, %s, %s, %s' % (vreg(dst), vreg(q), vreg(den), vreg(num)), vr(q) + vr(den) + vr(num), vr(dst), emu)


def v_mul(dst, a, b):
    """dst = a * b: a a VGPR number, ('s', n) or a float literal; b a VGPR number"""
    return B.v_f32_op('mul', dst, _vv(a) if isinstance(a, int) else a, _vv(b))


def v_add(dst, a, b):
    return B.v_f32_op('add', dst, _vv(a), _vv(b))


def v_sub(dst, a, b):
    return B.v_f32_op('sub', dst, _vv(a), _vv(b))


def _separate(ops):
    """the two adjacency rules of isa.check_hazards_stream hold inside the list whatever the scheduler puts in between: no reader
    directly behind a transcendental or behind a half-register write of the same register"""
    out = []
    for ins in ops:
        if out:
            prev = out[-1]
            touched = set(ins.rd) | (set(ins.wr) if ins.partial else set())
            if (prev.kind == 'trans' and set(prev.wr) & set(ins.rd)) or (prev.partial and set(prev.wr) & touched):
                out.append(s_nop(0))
        out.append(ins)
    return out


def _to_rev(x, rh, rl, tmp):
    """to_rev (r2l_device.h): rh = x * HI, rl = fma(x, LO, fma(x, HI, -rh))"""
    return [v_mul(rh, ('s', S_HI), x),
            B.v_fma_f32(tmp, _vv(x), ('s', S_HI), _vv(rh), neg_c=True),
            B.v_fma_f32(rl, _vv(x), ('s', S_LO), _vv(tmp))]


def _turn_fraction(RH, RL, g):
    """g = frac(x 2^l / 2 pi) in [-1/2, 1/2] from the scaled pair (trig_pow2: t = rh 2^l, u = t - rint(t), g = fma(rl, 2^l, u); the
    scalings by 2^l are exact, so rl 2^l + u rounds as the fma does)"""
    return [B.v_rndne_f32(g, _vv(RH)), v_sub(g, RH, g), v_add(g, RL, g)]


def v_mix16(dst, high, a, hsrc=None, hhalf=0):
    """half `high` of dst = f16(a * 16 - h), h = 0 or the f16 half `hhalf` of v[hsrc]: v_fma_mixlo/hi_f16 with f32 a, f32 16.0 (SGPR) and
    an f16 third source.  One instruction for split_store's `(f16)(v * act_scale)` (h = 0: the product by a power of two is exact,
    one rounding) and one for its residual `(f16)fma((float)h, -1, v * act_scale)` (exact before the rounding to f16)"""
    op = 'v_fma_mixhi_f16' if high else 'v_fma_mixlo_f16'
    if hsrc is None:
        text = '%s %s, %s, %s, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]' % (op, vreg(dst), vreg(a), sreg(S_A16))
    else:
        text = '%s %s, %s, %s, -%s op_sel:[0,0,%d] op_sel_hi:[0,0,1]' % (op, vreg(dst), vreg(a), sreg(S_A16), vreg(hsrc), hhalf)

    def emu(st):
        h = np.zeros(64) if hsrc is None else ((st.V[hsrc] >> (16 * hhalf)) & 0xffff).astype(np.uint16).view(np.float16).astype(np.float64)
        r = (st.f32('v', a).astype(np.float64) * ACT - h).astype(np.float32).astype(np.float16).view(np.uint16).astype(np.uint32)
        st.V[dst] = (st.V[dst] & 0x0000ffff) | (r << 16) if high else (st.V[dst] & 0xffff0000) | r
    return valu(text, vr(a) + (vr(hsrc) if hsrc is not None else []), vr(dst), emu, partial=True)


def _pack_pair(dst_hi, dst_lo, a0, a1, tmp=None):
    """split_store of two consecutive elements a0, a1 (before the factor act_scale): dword of the hi fragment = (f16(16 a0), f16(16 a1)),
    of the lo fragment their fp16 residuals.  tmp = (hi, lo) VGPR temporaries when the fragments live in AGPRs"""
    H, Lo = tmp if tmp else (dst_hi, dst_lo)
    ops = [v_mix16(H, 0, a0), v_mix16(Lo, 0, a0, H, 0), v_mix16(H, 1, a1), v_mix16(Lo, 1, a1, H, 1)]
    # (the lo half of Lo reads only the lo half of H, written two instructions earlier; _separate puts a nop where a reader would
    # directly follow a half-register write of the register it reads)
    if tmp:
        ops += [v_accw(dst_hi, H), v_accw(dst_lo, Lo)]
    return ops


def _element(BH, BL, scale, S1, S2, cos_mask):
    """S1 = sin or cos (lanes of cos_mask: cos; None: sin everywhere; 'all': cos everywhere) of 2 pi frac(scale (BH + BL)): trig_pow2 of
    r2l_device.h with pow2l = scale -- t = rh 2^l, u = t - rint(t), g = fma(rl, 2^l, u), cos(2 pi g) = sin(2 pi (1/4 - |g|))"""
    if scale == 1.0:
        ops = [B.v_rndne_f32(S2, _vv(BH)), v_sub(S1, BH, S2), v_add(S1, BL, S1)]
    else:
        ops = [v_mul(S1, float(scale), BH), B.v_rndne_f32(S2, _vv(S1)), v_sub(S1, S1, S2), B.v_fmac_lit(S1, float(scale), BL)]
    if cos_mask == 'all':
        ops += [v_sub_abs(S1, S_C25, S1)]
    elif cos_mask is not None:
        ops += [v_sub_abs(S2, S_C25, S1), v_cndmask(S1, S1, S2, cos_mask)]
    return ops


def _normalise_ops(d, N, W):
    """d[0..2] <- d / |d| (main.py:154-156) as hipcc expands sqrtf and __fdiv_rn for operands that need no scaling: v_sqrt_f32 and the
    neighbour that squares closest; v_div_scale x 2, v_rcp_f32, the Newton steps, v_div_fmas, v_div_fixup"""
    a, b, c2, r1, r2 = W
    ops = [v_mul(N, d[0], d[0]), v_mul(a, d[1], d[1]), v_add(N, N, a), v_mul(a, d[2], d[2]), v_add(N, N, a)]
    ops += [v_sqrt(a, N), v_addi(b, -1, a), v_addi(c2, 1, a), v_fma_na(r1, b, a, N), v_fma_na(r2, c2, a, N),
            v_cmp0('ge', S_TM, r1), s_nop(1), v_cndmask(a, a, b, S_TM), v_cmp0('lt', S_TM, r2), s_nop(1), v_cndmask(N, a, c2, S_TM)]
    DS, NS, R, Q, E = W
    for k in range(3):
        ops += [v_div_scale(DS, S_TM, N, N, d[k]), v_rcp(R, DS), v_div_scale(NS, 'vcc', d[k], N, d[k]), v_fma_na(E, DS, R, 1.0), v_fmac(R, E, R),
                v_mul(Q, NS, R), v_fma_na(E, DS, Q, NS), v_fmac(Q, E, R), v_fma_na(E, DS, Q, NS), v_div_fmas(Q, E, R, Q), v_div_fixup(d[k], Q, N, d[k])]
    return ops


def emb_E_ops(c):
    """E fragments (both k-steps, hi | lo) of column tile c from its (o, d, z) in the RAW AGPRs: nerf_tile_embed's E part; then the
    view direction d / |d| back into d's RAW AGPRs (emb_V_ops reads it there: the normalisation runs in this, the longer, window)"""
    T = [V_OUT + i for i in range(16)]
    xs, rh, rl = T[0:3], [T[3], T[5], T[7]], [T[4], T[6], T[8]]
    BH, BL, S1, S2, A, Hh = T[9], T[10], T[11], T[12], [T[13], T[14]], T[15]
    ops = [v_accr(BH, RAW_Z(c))]
    for k in range(3):
        ops += [v_accr(S1, RAW_O(c) + k), v_accr(S2, RAW_D(c) + k), v_mul(S2, S2, BH), v_add(xs[k], S1, S2)]     # o + d * z  (main.py:701)
    for k in range(3):
        ops += _to_rev(xs[k], rh[k], rl[k], S1)
    # k-step 0: coordinate q >> 1, frequencies 0..7; odd lane quarters hold the cosines
    ops += [v_cndmask(BH, rh[0], rh[1], S_MQ2), v_cndmask(BL, rl[0], rl[1], S_MQ2)]
    for j in range(8):
        ops += _element(BH, BL, 2.0 ** j, S1, S2, S_MQ1) + [B.v_sin_f32(A[j & 1], _vv(S1))]
        ops[-1].cost = 4
        if j & 1:
            ops += _pack_pair(E_reg('E', 0, c, False) + (j >> 1), E_reg('E', 0, c, True) + (j >> 1), A[0], A[1], (Hh, S2))
    # k-step 1: lane quarters 0, 1: coordinate 2, frequencies 0..7; quarters 2, 3: frequencies 8, 9 of all three coordinates, then
    # the identity.  Per element pair p one base pair: r2 in quarters 0, 1 and r_p 2^(8 - 2p) in quarters 2, 3, scaled by 4^p 2^e
    for p in range(4):
        if p < 3:
            ops += [v_mul(S1, float(2 ** (8 - 2 * p)), rh[p]), v_cndmask(BH, rh[2], S1, S_MQ2),
                    v_mul(S1, float(2 ** (8 - 2 * p)), rl[p]), v_cndmask(BL, rl[2], S1, S_MQ2)]
            bh, bl = BH, BL
        else:
            bh, bl = rh[2], rl[2]
        for e in range(2):
            ops += _element(bh, bl, 4.0 ** p * 2.0 ** e, S1, S2, S_MQ1) + [B.v_sin_f32(A[e], _vv(S1))]
            ops[-1].cost = 4
            if p == 3:      # j = 6: q == 3 ? xs[2] : xs[0];  j = 7: q == 3 ? 0 : xs[1]   (quarters 2, 3 only)
                ops += [v_cndmask(S2, xs[0], xs[2], S_MQ3) if e == 0 else v_cndmask(S2, xs[1], None, S_MQ3), v_cndmask(A[e], A[e], S2, S_MQ2)]
        ops += _pack_pair(E_reg('E', 1, c, False) + p, E_reg('E', 1, c, True) + p, A[0], A[1], (Hh, S2))
    d = T[0:3]
    ops += [v_accr(d[k], RAW_D(c) + k) for k in range(3)]
    ops += _normalise_ops(d, T[3], T[4:9])
    ops += [v_accw(RAW_D(c) + k, d[k]) for k in range(3)]
    return _separate(ops)


def V_STAGE(c, lo):
    """where the next tile's view fragments wait for the V layer's last embedding k-step: k-steps 4 (hi) and 5 (lo) of set P,
    dead between FA's last read of P and L1's epilogue of the next block"""
    return hset('P', 5 if lo else 4, c)


def emb_V_ops(c):
    """view fragments (hi | lo) of column tile c from d / |d| (left in d's RAW AGPRs by emb_E_ops): nerf_tile_embed's view step, into
    V_STAGE.  Lane quarter q < 3: component q, elements j = 0..3 sin, 4..7 cos of frequencies 0..3; q = 3: the identity columns"""
    T = [V_OUT + i for i in range(16)]
    vs, BH, BL, S1, S2, As, Ac = T[0:3], T[9], T[10], T[11], T[12], [T[13], T[14]], [T[4], T[5]]
    ops = [v_accr(vs[k], RAW_D(c) + k) for k in range(3)]
    ops += [v_cndmask(S1, vs[2], vs[1], S_MQ1E), v_cndmask(S1, S1, vs[0], S_MQ0)]
    ops += _to_rev(S1, BH, BL, S2)
    for f in range(4):
        ops += _element(BH, BL, 2.0 ** f, S1, S2, None)                     # g of frequency f
        ops += [v_sub_abs(S2, S_C25, S1), v_sin(As[f & 1], S1), v_sin(Ac[f & 1], S2),
                v_cndmask(As[f & 1], As[f & 1], vs[f] if f < 3 else None, S_MQ3), v_cndmask(Ac[f & 1], Ac[f & 1], None, S_MQ3)]
        if f & 1:
            r = f >> 1
            ops += _pack_pair(V_STAGE(c, False) + r, V_STAGE(c, True) + r, As[0], As[1])            # elements 2 r, 2 r + 1: sines
            ops += _pack_pair(V_STAGE(c, False) + 2 + r, V_STAGE(c, True) + 2 + r, Ac[0], Ac[1])    # elements 4 + 2 r, ...: cosines
    return _separate(ops)


def emb_V_commit_ops():
    ops = []
    for c in range(NC):
        for lo in (False, True):
            for r in range(4):
                ops.append(v_accw(E_reg('V', 0, c, lo) + r, V_STAGE(c, lo) + r))
    return ops


# ---- the ray loads of a tile and the raw stores: address arithmetic in VALU, global_load into the RAW AGPRs -------------------------
def _vop(text, rd, wr, emu):
    return valu(text, rd, wr, emu)


def _u32(st, x):
    return st.V[x].astype(np.uint64) if isinstance(x, int) else np.full(64, np.uint64(st.S[x[1]]), dtype=np.uint64)


def v_u32(op, dst, a, b):
    """dst = a op b on 32-bit unsigned lanes; a: VGPR number, ('s', n) or a small int; b: VGPR number (VOP2 operand order: the
    reverse forms are named explicitly)"""
    names = {'add': 'v_add_u32_e32', 'sub': 'v_sub_u32_e32', 'min': 'v_min_u32_e32', 'lshr': 'v_lshrrev_b32_e32', 'lshl': 'v_lshlrev_b32_e32',
             'and': 'v_and_b32_e32', 'mul_lo': 'v_mul_lo_u32', 'mul_hi': 'v_mul_hi_u32'}

    def val(st, x):
        if isinstance(x, int) and not isinstance(x, bool):
            return st.V[x].astype(np.uint64)
        if isinstance(x, tuple) and x[0] == 's':
            return np.full(64, np.uint64(int(st.S[x[1]]) & 0xffffffff), dtype=np.uint64)
        return np.full(64, np.uint64(x[1]), dtype=np.uint64)          # ('i', value)

    def emu(st):
        x, y = val(st, a), val(st, b)
        if op == 'add':
            r = x + y
        elif op == 'sub':
            r = x - y
        elif op == 'min':
            r = np.minimum(x, y)
        elif op == 'lshr':        # b >> a  (rev form)
            r = y >> (x & np.uint64(31))
        elif op == 'lshl':
            r = y << (x & np.uint64(31))
        elif op == 'and':
            r = x & y
        elif op == 'mul_lo':
            r = x * y
        else:
            r = (x * y) >> np.uint64(32)
        st.V[dst] = (r & np.uint64(0xffffffff)).astype(np.uint32)

    def txt(x):
        if isinstance(x, int) and not isinstance(x, bool):
            return vreg(x)
        return sreg(x[1]) if x[0] == 's' else str(x[1])
    return valu('%s %s, %s, %s' % (names[op], vreg(dst), txt(a), txt(b)), [('v', x) for x in (a, b) if isinstance(x, int) and not isinstance(x, bool)],
                vr(dst), emu)


def gload(adst, n, voff, sbase, buf):
    """a[adst : adst + n] <- n dwords per lane from the global buffer `buf` at byte offset v[voff] (saddr form, vmcnt)"""
    text = 'global_load_dword%s %s, %s, %s' % ({1: '', 3: 'x3'}[n], areg(adst, n), vreg(voff), sreg(sbase, 2))

    def emu(st):
        g = st.gmem[buf]
        off = st.V[voff].astype(np.int64)
        data = np.zeros((n, 64), dtype=np.uint32)
        for l in range(64):
            assert off[l] % 4 == 0 and 0 <= off[l] and off[l] + 4 * n <= g.nbytes, (buf, int(off[l]), g.nbytes)
            data[:, l] = g.view(np.uint32).reshape(-1)[off[l] // 4:off[l] // 4 + n]
        st.vmq.append(('v', ('a', adst, data)))
        st.pend_regs.update(ar(adst, n))
    return Ins(text, 'vload', rd=vr(voff), wr=ar(adst, n), emu=emu)


def raw_load_ops():
    """(o, d, z) of the lane's four points of tile s[S_NEXT] into the RAW AGPRs: nerf_tile_load (csrc/nerf_kernels.hip) for NC = 4
    without given view directions.  pt / S by the round-up multiply of Granlund & Montgomery (host: magic, sh1, sh2)"""
    G0, G1, G2, G3 = V_G, V_G + 1, V_G + 2, V_G + 3
    ops = [salu('s_lshl_b32 %s, %s, 8' % (sreg(S_TB), sreg(S_NEXT)), lambda st: st.S.__setitem__(S_TB, (st.S[S_NEXT] << 8) & 0xffffffff))]
    for c in range(NC):
        ops += [v_u32('add', G0, ('s', S_TB), V_PL)]
        if c:
            ops += [v_u32('add', G0, ('i', 16 * c), G0)]
        ops += [v_u32('min', G0, ('s', S_LAST), G0),                 # tail lanes repeat the last point (their stores are masked)
                v_u32('mul_hi', G1, G0, ('s', S_MAGIC)), v_u32('sub', G2, G0, G1), v_u32('lshr', G2, ('s', S_SH1), G2), v_u32('add', G2, G1, G2),
                v_u32('lshr', G2, ('s', S_SH2), G2),                 # ray = pt / S
                v_u32('mul_lo', G1, G2, ('s', S_S)), v_u32('sub', G1, G0, G1),           # sample = pt - ray * S
                v_u32('mul_lo', G3, G2, ('s', S_ZS)), v_u32('add', G3, G3, G1), v_u32('lshl', G3, ('i', 2), G3),
                v_u32('mul_lo', G2, G2, ('i', 12)),
                gload(RAW_O(c), 3, G2, S_PO, 'rays_o'), gload(RAW_D(c), 3, G2, S_PD, 'rays_d'), gload(RAW_Z(c), 1, G3, S_PZ, 'z')]
    return ops


def raw_store_ops():
    """raw[pt] = (rgb, sigma) / act_scale of tile s[S_TILE] for lanes 0..15 and pt < n_pts (nerf_chain_kernel's store)"""
    G0, G1 = V_G, V_G + 1
    ops = [salu('s_lshl_b32 %s, %s, 8' % (sreg(S_TB), sreg(S_TILE)), lambda st: st.S.__setitem__(S_TB, (st.S[S_TILE] << 8) & 0xffffffff))]
    for c in range(NC):
        ops.append(B.v_mov_b32(V_OUT + 4 * c + 3, _vv(V_SIG + c)))
        ops += [v_mul(V_OUT + 4 * c + k, 1.0 / ACT, V_OUT + 4 * c + k) for k in range(4)]
        ops += [v_u32('add', G0, ('s', S_TB), V_PL)]
        if c:
            ops += [v_u32('add', G0, ('i', 16 * c), G0)]
        ops += [v_u32('lshl', G1, ('i', 4), G0)]

        def emu_cmp(st, G0=G0):
            st.vcc = st.V[G0].astype(np.int64) < int(st.S[S_NPTS])
        ops.append(valu('v_cmp_gt_u32_e32 vcc, %s, %s' % (sreg(S_NPTS), vreg(G0)), vr(G0), [], emu_cmp))

        def emu_exec(st):
            st.exec_ = st.vcc & st.S[S_ML16]
        ops.append(salu('s_and_b64 exec, vcc, %s' % sreg(S_ML16, 2), emu_exec))

        def emu_store(st, c=c, G1=G1):
            g = st.gmem['raw'].view(np.uint32).reshape(-1)
            off = st.V[G1].astype(np.int64)
            for l in range(64):
                if st.exec_[l]:
                    assert off[l] % 16 == 0 and 0 <= off[l] and off[l] + 16 <= g.nbytes, (int(off[l]), g.nbytes)
                    g[off[l] // 4:off[l] // 4 + 4] = st.V[V_OUT + 4 * c:V_OUT + 4 * c + 4, l]
            st.vmq.append(('st', None))
        ops.append(Ins('global_store_dwordx4 %s, %s, %s' % (vreg(G1), vreg(V_OUT + 4 * c, 4), sreg(S_PRAW, 2)), 'vstore',
                       rd=vr(G1) + vr(V_OUT + 4 * c, 4), emu=emu_store))
        ops.append(salu('s_mov_b64 exec, -1', lambda st: setattr(st, 'exec_', np.ones(64, bool))))
    return ops


def vm_wait(n):
    """s_waitcnt vmcnt(n) over the ONE in-order queue of this stream's vector memory operations (LDS-DMA pieces, register loads,
    stores): all but the youngest n have completed -- register loads land, and the LDS-DMA pieces among the youngest n are what
    the next barrier must not certify (isa.barrier: st.cert)"""
    def emu(st):
        q = st.vmq
        done, st.vmq = (q[:len(q) - n], q[len(q) - n:]) if n else (q, [])
        for kind, payload in done:
            if kind == 'v':
                file, dst, data = payload
                st.regs(file)[dst:dst + len(data)] = data
                for r in (vr if file == 'v' else ar)(dst, len(data)):
                    st.pend_regs.discard(r)
        st.cert = sum(1 for kind, _ in st.vmq if kind == 'd')
    return Ins('s_waitcnt vmcnt(%d)' % n, 'wait', emu=emu)



class Opts:
    def __init__(self, **kw):
        self.lead = 8          # anchors a fragment read is issued ahead of its first MFMA
        self.lead6 = 5
        self.cap = 3           # issue slots for fillers behind each MFMA
        self.dma_gap = 4       # anchors (16-cycle MFMAs) between two LDS-DMA pieces: 4 .. 6 are 1.7 % faster than 3, 2 is slower
        self.pair = False
        self.wait_group = 2    # fp16 fragments one s_waitcnt may cover (those already issued)
        if P3:                 # same-box sweep (gpurun_out/teacher_variants.log, round 5): cap 3 / gap 4 97.6 ms per frame, cap 2 98.2, gap 6 97.3,
            self.cap = 2       # cap 2 + gap 6 96.6, + lead 12 97.4, + wait group 3 98.2
            self.dma_gap = 6
        self.cap_late = 2      # f16c4e: issue slots behind an MFMA once the embedding fillers may run (anchor >= a_emb)
        if NC > 2:             # three (four) MFMAs per fragment: fewer filler slots behind each, LDS-DMA pieces further apart (same-box sweep:
            self.cap = 2       # cap 2 / 3 / 4 / 5 = 36.9 / 37.2 / 37.3 / 37.6 ms per frame, gap 3 / 4 / 6 = 37.5 / 37.2 / 36.9, both: -1.4 %)
            self.dma_gap = 2 * NC          # (four column tiles: gap 4 / 6 / 8 = 34.75 / 34.35 / 34.2 ms)
        self.__dict__.update(kw)


def lds_addr(slot, byte_off, width):
    off = slot * SLOT + byte_off
    lo, hi = (V_L0, V_L1) if width == 16 else (V_L8A, V_L8B)
    return (lo, off) if off < 65536 else (hi, off - 65536)


def rdv_anchor(ci):
    """anchor in front of which chunk ci's rendezvous (vmcnt wait + barrier + refill) sits: the middle of the chunk"""
    ts = CHUNKS[ci]['tiles']
    if len(ts) >= 2:
        return afirst(ts[len(ts) // 2])
    return afirst(ts[0]) + len(ANCH[ts[0]]) // 2


def chunk_issue_seq(cn, slot):
    """SALU + LDS-DMA instructions that fetch chunk cn (index into the tile's stream, wrapped) into ring slot `slot`"""
    ch = CHUNKS[cn % NCH]
    pw = ch['pw']
    off = CHUNK_OFF[cn % NCH]
    wp = S_WPW + pw - 1
    seq = [salu('s_add_u32 %s, %s, %s' % (sreg(S_G), sreg(S_W), sreg(wp)), lambda st: st.S.__setitem__(S_G, st.S[S_W] + st.S[wp])),
           salu('s_addc_u32 %s, %s, 0' % (sreg(S_G + 1), sreg(S_W + 1)))]
    if off:
        seq += [salu('s_add_u32 %s, %s, 0x%x' % (sreg(S_G), sreg(S_G), off), lambda st: st.S.__setitem__(S_G, st.S[S_G] + off)),
                salu('s_addc_u32 %s, %s, 0' % (sreg(S_G + 1), sreg(S_G + 1)))]
    seq.append(salu('s_add_u32 m0, %s, 0x%x' % (sreg(wp), slot * SLOT), lambda st: setattr(st, 'm0', st.S[wp] + slot * SLOT)))
    seq.append(s_nop(0))
    for i in range(pw):
        if i == 4:
            seq.append(salu('s_add_u32 m0, m0, 0x1000', lambda st: setattr(st, 'm0', st.m0 + 4096)))
            seq.append(s_nop(0))
        seq.append(dma_piece(i, pw, tag=('dma', cn, i)))
    return seq, pw


def build_fillers(opts):
    F = []
    bufmap = {}
    # ---- operands in consumption order -------------------------------------------------------------
    first, last, order = {}, {}, []
    for T in range(NT):
        for i, (kind, k, c, p) in enumerate(ANCH[T]):
            key = operand_key(T, kind, k, p)
            a = ABASE[T] + i
            if key not in first:
                first[key] = a
                order.append(key)
            last[key] = a
    hi_keys = [k for k in order if k[0] != 'a6']
    a6_keys = [k for k in order if k[0] == 'a6']

    def certified(T):
        """anchor behind which the chunk of tile T may be read"""
        ci, _ = TILE_CHUNK[T]
        return -1 if ci < 3 else rdv_anchor(ci - 1) + 1

    for n, key in enumerate(hi_keys):
        what, T, k = key
        L = TILES[T].layer
        ci, kc = TILE_CHUNK[T]
        pc, po = piece_of(L, kc, what, k)
        bv, off = lds_addr(ci % NSLOT, pc * 1024 + po, 16)
        buf = V_HI + (n % N_FRAG_BUF) * 4
        bufmap[key] = buf
        hard = last[hi_keys[n - N_FRAG_BUF]] if n >= N_FRAG_BUF else -1
        F.append(Filler(ds_read_b128(buf, bv, off, tag=key), max(hard, first[key] - opts.lead, certified(T)), first[key], ('rd',)))
    for n, key in enumerate(a6_keys):
        what, T, k = key
        L = TILES[T].layer
        ci, kc = TILE_CHUNK[T]
        buf = V_A6 + (n % 2) * 6
        bufmap[key] = buf
        hard = last[a6_keys[n - 2]] if n >= 2 else -1
        e = max(hard, first[key] - opts.lead6, certified(T))
        pc, po = piece_of(L, kc, 'b6', k)
        bv, off = lds_addr(ci % NSLOT, pc * 1024 + po, 16)
        F.append(Filler(ds_read_b128(buf, bv, off, tag=key + (0,)), e, first[key], ('rd6',)))
        pc, po = piece_of(L, kc, 'b6b', k)
        bv, off = lds_addr(ci % NSLOT, pc * 1024 + po, 8)
        F.append(Filler(ds_read_b64(buf + 4, bv, off, tag=key + (1,)), e, first[key], ('rd6',)))
    # ---- bias and scales (resident table) ------------------------------------------------------------
    for T, t in enumerate(TILES):
        F.append(Filler(ds_read_b128(V_BIAS + (T & 1) * 4, V_AUX, t.li * AUX_LAYER + 64 * t.u, tag=('bias', T)),
                        afirst(T - 1) + 1 if T >= 1 else -1, afirst(T), ('aux',)))
        if t.u == 0 and t.layer.nj:
            prev0 = TILE_OF[(t.li - 1, 0)]      # layer li-2 (same scale registers) is over once layer li-1 runs
            e = afirst(prev0) + 1
            if t.li >= 2 and CHAIN[t.li - 2].uses_inv:      # ... unless it multiplies by them in its epilogues: those end under tile 0 of layer li-1
                e = afirst(prev0 + 1) + 1
            F.append(Filler(ds_read_b64(V_SC + 2 * (t.li & 1), V_AUX, t.li * AUX_LAYER + AUX_SCALES, tag=('scale', t.li)),
                            e, aidx(T, 'm6', 0, 0), ('aux',)))
        if t.u == 0 and t.layer.uses_inv:
            # 1 / weight scale of layer li for its epilogues (they run under the layer's tiles 1.. and under the next layer's tile 0);
            # the registers' previous owner, layer li - 2, had its last epilogue under tile 0 of layer li - 1
            e = afirst(TILE_OF[(t.li - 1, 0)] + 1) + 1 if t.li >= 1 else -1
            F.append(Filler(ds_read_b64(V_SC + 2 * (t.li & 1), V_AUX, t.li * AUX_LAYER + AUX_SCALES, tag=('scale', t.li)),
                            e, afirst(T) + 1, ('aux',)))
    # ---- epilogue of tile T-1 under tile T -------------------------------------------------------------
    for T in range(1, NT):
        tp = TILES[T - 1]
        if SKIPV and tp.layer.name == 'A':
            continue                        # the density's epilogue runs AT the cut (cut_ops), in front of the decision
        for c in range(NC):
            e0 = afirst(T) + 2 + c          # two further MFMAs behind the last writer of its accumulator
            for ins, cons in epilogue_ops(T - 1, c):
                dl = afirst(T + 1)          # the accumulator buffer is reused by tile T+1
                if cons is not None:
                    Ln = CHAIN[tp.li + 1]
                    T0n = TILE_OF[(tp.li + 1, 0)]
                    if cons[0] == 'hi':
                        if cons[1] < Ln.ks:
                            dl = min(dl, aidx(T0n, 'm16', cons[1], 0) - 2)
                    elif cons[0] == 'lo':
                        if cons[1] < Ln.ks:
                            dl = min(dl, aidx(T0n, 'm16', cons[1], 0, 1) - 2)
                    elif (cons[1], cons[2]) in Ln.j_order():
                        dl = min(dl, aidx(T0n, 'm6', Ln.j_order().index((cons[1], cons[2])), 0) - 2)
                F.append(Filler(ins, e0, dl, ('epi', c)))
    # ---- rendezvous + refill -----------------------------------------------------------------------------
    group = {}                          # chunk number -> LDS-DMA instructions per wave
    for ci in range(NCH):
        ar = rdv_anchor(ci)
        nxt = rdv_anchor(ci + 1) if ci + 1 < NCH else N_ANCH
        ch = ('dma',)
        F.append(Filler((vm_wait if EMB else waitcnt_vm)(group.get(ci + 2, 0)), ar - 1, ar + 1, ch))
        F.append(Filler(barrier(), ar - 1, ar + 1, ch))
        seq, pw = chunk_issue_seq(ci + 3, (ci + 3) % NSLOT)
        group[ci + 3] = pw
        k = 0
        for ins in seq:
            if ins.kind != 'dma' and k == 0:
                F.append(Filler(ins, ar - 1, min(ar + 4, nxt - 1), ch))
            else:
                F.append(Filler(ins, min(ar + 1 + opts.dma_gap * k, nxt - 2), nxt - 1, ch))
                if ins.kind == 'dma':
                    k += 1
    if EMB:
        # ---- the next tile's rays: 12 global loads into the RAW AGPRs, two instructions per anchor from the start of L0 ------------
        # (their destinations were read for the last time by the previous block's embedding fillers)
        for i, ins in enumerate(raw_load_ops()):
            F.append(Filler(ins, i // 2, afirst(TILE_OF[(1, 0)]), ('raw',)))
        # ---- the next tile's embedding: lowest priority (deadline = the end of the block), in the shadow of L6 .. RGB ---------------
        TL5 = TILE_OF[(5, CHAIN[5].rt - 1)]
        a_e = max(aidx(TL5, 'x', xi, c, p) for xi in range(CHAIN[5].nx) for c in range(NC) for p in range(XPASS)) + 2    # E was read for the last time
        TFA = TILE_OF[(8, CHAIN[8].rt - 1)]
        a_v = max(aidx(TFA, 'm16', s_, c, 0) for s_ in range(4, 8) for c in range(NC)) + 2         # FA was the last reader of P's upper half
        TV = TILE_OF[(9, CHAIN[9].rt - 1)]
        a_c = max(aidx(TV, 'x', 0, c, p) for c in range(NC) for p in range(XPASS)) + 2             # the V layer's last embedding k-step
        ch = ('emb',)
        for c in range(NC):
            for ins in emb_E_ops(c):
                F.append(Filler(ins, a_e, N_ANCH + 1, ch))
        for c in range(NC):
            for ins in emb_V_ops(c):
                F.append(Filler(ins, a_v, N_ANCH + 1, ch))
        for ins in emb_V_commit_ops():
            F.append(Filler(ins, a_c, N_ANCH + 1, ch))
        opts.a_emb = a_e
    return F, bufmap


def cut_ops(T_A):
    """f16p3s / mixs: behind the last MFMA of layer A (tile T_A) -- its epilogue (sigma into the output operands), then the decision: does any
    of the workgroup's 128 points have a positive density?  Row 0 of the tile's accumulators is sigma x act_scale (x 2^k): lanes 0..15
    of register 0 (the other rows of the tile have zero weights and zero bias: never > 0).  The wave's answer goes into this tile's LDS
    word (%[fl]: the HIP code alternates between two words, the other one is cleared here for the next tile: every wave has read it
    before this tile's entry barrier), one barrier, every wave reads the OR back and takes the same exit."""
    acc = [ACC(T_A & 1, c) for c in range(NC)]
    t0 = TMP(0)
    v_any, v_addr, v_addr2, v_zero = t0 + 6, t0 + 7, t0 + 8, t0 + 9
    m0s, m1s, sf = S_CUT, S_CUT + 2, S_CUT + 2
    ops = [s_nop(15), s_nop(15)]
    if TILES[T_A].layer.uses_inv:
        ops.append(('need', ('scale', TILES[T_A].li)))
    for c in range(NC):
        ops += [ins for ins, _ in epilogue_ops(T_A, c)]

    def decide(st):
        anyp = False
        for c in range(NC):
            anyp = anyp or bool((st.f32('v', acc[c]) > 0).any())
        st.skip = not anyp
    ops.append(valu('v_cmp_lt_f32_e64 %s, 0, %s' % (sreg(m0s, 2), vreg(acc[0])), vr(acc[0]), [], decide))
    ops.append(valu('v_cmp_lt_f32_e64 %s, 0, %s' % (sreg(m1s, 2), vreg(acc[1])), vr(acc[1]), [], None))
    ops.append(s_nop(4))
    ops.append(salu('s_or_b64 %s, %s, %s' % (sreg(m0s, 2), sreg(m0s, 2), sreg(m1s, 2))))
    ops.append(salu('s_cselect_b32 %s, 1, 0' % sreg(sf)))
    ops.append(valu('v_mov_b32 %s, %s' % (vreg(v_any), sreg(sf)), [], vr(v_any), None))
    ops.append(valu('v_mov_b32 %s, %%[fl]' % vreg(v_addr), [], vr(v_addr), None))
    ops.append(valu('v_xor_b32 %s, 4, %s' % (vreg(v_addr2), vreg(v_addr)), vr(v_addr), vr(v_addr2), None))
    ops.append(valu('v_mov_b32 %s, 0' % vreg(v_zero), [], vr(v_zero), None))
    ops.append(Ins('ds_write_b32 %s, %s' % (vreg(v_addr2), vreg(v_zero)), 'salu'))       # (kind: outside the scheduler's LDS-read accounting;
    ops.append(Ins('ds_max_u32 %s, %s' % (vreg(v_addr), vreg(v_any)), 'salu'))           #  the waits around them are lgkmcnt(0))
    ops.append(waitcnt_lgkm(0))
    ops.append(barrier())
    ops.append(Ins('ds_read_b32 %s, %s' % (vreg(v_any), vreg(v_addr)), 'salu'))
    ops.append(waitcnt_lgkm(0))
    ops.append(salu('v_readfirstlane_b32 %s, %s' % (sreg(sf), vreg(v_any))))
    ops.append(salu('s_cmp_eq_u32 %s, 0' % sreg(sf)))
    ops.append(Ins('s_cbranch_scc1 L_skip_%=', 'salu', tag='cut'))
    return ops


def skip_tail_ops():
    """the second exit: nothing of the feature rows / views / rgb layers runs.  LDS-DMA pieces of chunks behind the cut are in flight and
    fragment reads may be: wait for both, one barrier (every wave's pieces have landed, nobody reads a slot any more), then the next tile's
    chunks 0..2 into slots 0..2 as the full path's last refills do, and zeros for the colours"""
    ops = [waitcnt_vm(0), waitcnt_lgkm(0), barrier()] + prologue_ops()
    for c in range(NC):
        ops += [v_zero_out(c * 4 + k) for k in range(3)]
    return ops


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
        """wait for the LDS read `key`; reads in `also` that have been issued ride along (one s_waitcnt for a run of MFMAs)"""
        idx = self.ds_index[key]
        if idx < self.ds_done:
            return
        for k2 in also:
            if k2 in self.ds_index:
                idx = max(idx, self.ds_index[k2])
        n = min(15, self.ds_issued - idx - 1)
        self.emit(waitcnt_lgkm(n))
        self.ds_done = self.ds_issued - n


def schedule(opts):
    """the tile block between its entry (vmcnt(0) + barrier) and its exposed last epilogue: [Ins]"""
    sch = Sched()
    fillers, bufmap = build_fillers(opts)
    for i, f in enumerate(fillers):
        f.seq = i
    chains = {}
    for f in fillers:
        chains.setdefault(f.chain, []).append(f)
    for ch in chains.values():
        for i in range(len(ch) - 2, -1, -1):   # a filler must not hold up a successor with an earlier deadline
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
    T = 0
    a_cut = afirst(TILE_OF[([l.name for l in CHAIN].index('F'), 0)]) if SKIPV else -1
    for a in range(N_ANCH):
        while True:
            r = [f for f in ready(a - 1) if f.deadline <= a]
            if not r:
                break
            issue(r[0])
        while a >= ABASE[T + 1]:
            T += 1
        if a == a_cut:          # the second exit sits between layer A's last MFMA and layer F's first
            for ins in cut_ops(T - 1):
                if isinstance(ins, tuple):
                    sch.need(ins[1])
                else:
                    sch.emit(ins)
                    if ins.kind == 'wait' and 'lgkmcnt(0)' in ins.text:
                        sch.ds_done = sch.ds_issued
        t = TILES[T]
        L = t.layer
        kind, k, c, p = ANCH[T][a - ABASE[T]]
        d = ACC(T & 1, c)
        is_first = (a - ABASE[T]) == c           # the first MFMA of column tile c starts from the bias
        csrc = V_BIAS + (T & 1) * 4 if is_first else d
        if is_first:
            sch.need(('bias', T))
        if a == ABASE[T] and T >= 1 and TILES[T - 1].layer.uses_inv:
            sch.need(('scale', TILES[T - 1].li))       # the epilogue of tile T - 1 (under this tile) multiplies by its layer's 1 / weight scale
        key = operand_key(T, kind, k, p)
        if kind == 'x':
            sch.need(key)
            ek, e = L.extra[k]
            ins = mfma16(d, bufmap[key], 'a', E_reg(ek, e, c, p == 1), csrc, tag=('x', T, k, c, p),
                         btext=None if EMB else E_name(ek, e, c, p == 1))
        elif kind == 'm16':
            sch.need(key, [('hi', T, k + g) for g in range(1, opts.wait_group)])
            if p == 1:        # f16p3: hi(W) x lo(a)
                ins = mfma16(d, bufmap[key], 'a', lset(L.src, k, c), csrc, tag=('m16', T, k, c, p))
            else:
                ins = mfma16(d, bufmap[key], hfile(L.src), hset(L.src, k, c), csrc, tag=('m16', T, k, c) + ((p,) if p else ()))
        else:
            sch.need(key + (1,))
            term, tt = L.j_order()[k]
            if t.u == 0 and k == 0 and c == 0:
                sch.need(('scale', t.li))
            ins = mfma6(d, bufmap[key], b6(L.src, term, tt, c), V_SC + 2 * (t.li & 1) + term, V_SBA if term == 0 else V_SBL,
                        tag=('m6', T, k, c))
        sch.emit(ins)
        budget = opts.cap if not opts.pair else (0 if c == 0 else 2 * opts.cap)   # pair: fillers only behind the c = 1 MFMA
        if EMB and a >= opts.a_emb:
            budget = opts.cap_late
        while budget > 0:
            r = ready(a)
            if not r:
                break
            issue(r[0])
            budget -= r[0].ins.cost
    while True:          # what is left of the refill of the next tile's first chunks
        r = ready(N_ANCH + 10 ** 6)
        if not r:
            break
        issue(r[0])
    return sch.out


def setup_ops():
    """(text lines, emulator function) of the per-block constant setup"""
    L = []
    a = L.append
    a('s_mov_b32 %s, m0' % sreg(S_M0SAVE))
    a('s_mov_b64 %s, %%[wimg]' % sreg(S_W, 2))
    a('s_mov_b32 %s, %%[wave]' % sreg(S_WAVE))
    a('s_mov_b32 %s, 0xbf800000' % sreg(S_NEG1))
    a('s_lshl_b32 %s, %s, 10' % (sreg(S_WPW), sreg(S_WAVE)))
    for k in range(1, 8):
        a('s_add_u32 %s, %s, %s' % (sreg(S_WPW + k), sreg(S_WPW + k - 1), sreg(S_WPW)))
    a('v_mbcnt_lo_u32_b32 %s, -1, 0' % vreg(V_LANE))
    a('v_mbcnt_hi_u32_b32 %s, -1, %s' % (vreg(V_LANE), vreg(V_LANE)))
    a('v_lshlrev_b32 %s, 4, %s' % (vreg(V_L0), vreg(V_LANE)))
    a('v_add_u32 %s, 0x10000, %s' % (vreg(V_L1), vreg(V_L0)))
    a('v_add_u32 %s, 0x1000, %s' % (vreg(V_LOFF), vreg(V_L0)))
    a('v_lshlrev_b32 %s, 3, %s' % (vreg(V_L8A), vreg(V_LANE)))
    a('v_add_u32 %s, 0x10000, %s' % (vreg(V_L8B), vreg(V_L8A)))
    a('v_lshrrev_b32 %s, 4, %s' % (vreg(V_AUX), vreg(V_LANE)))
    a('v_lshlrev_b32 %s, 4, %s' % (vreg(V_AUX), vreg(V_AUX)))
    a('v_add_u32 %s, 0x%x, %s' % (vreg(V_AUX), LDS_AUX, vreg(V_AUX)))
    if V_SBA is not None and not X1:
        a('v_mov_b32 %s, 0x%08x' % (vreg(V_SBA), 0x01010101 * (127 + ACT_EXP)))
        a('v_mov_b32 %s, 0x%08x' % (vreg(V_SBL), 0x01010101 * (127 + RES_EXP)))
        a('v_mov_b32 %s, 0x%08x' % (vreg(V_CVA), f32_bits(2.0 ** ACT_EXP)))
        a('v_mov_b32 %s, 0x%08x' % (vreg(V_CVL), f32_bits(2.0 ** RES_EXP)))

    def emu(st):
        lanes = np.arange(64, dtype=np.uint32)
        st.V[V_LANE] = lanes
        st.V[V_L0] = lanes * 16
        st.V[V_L1] = lanes * 16 + 65536
        st.V[V_LOFF] = lanes * 16 + 4096
        st.V[V_L8A] = lanes * 8
        st.V[V_L8B] = lanes * 8 + 65536
        st.V[V_AUX] = LDS_AUX + (lanes >> 4) * 16
        if V_SBA is not None and not X1:
            st.V[V_SBA] = 0x01010101 * (127 + ACT_EXP)
            st.V[V_SBL] = 0x01010101 * (127 + RES_EXP)
            st.V[V_CVA] = f32_bits(2.0 ** ACT_EXP)
            st.V[V_CVL] = f32_bits(2.0 ** RES_EXP)
        st.S[S_W] = 0
        for k in range(8):
            st.S[S_WPW + k] = st.wave * (k + 1) * 1024
    return L, emu


def prologue_ops():
    """ring prologue: chunks 0, 1, 2 -> slots 0, 1, 2"""
    seq = []
    for k in range(3):
        seq += chunk_issue_seq(k, k)[0]
    return seq


def tail_ops():
    """exposed epilogue of the last row tile (RGB)"""
    ops = [s_nop(15), s_nop(15)]
    if CHAIN[-1].uses_inv:
        ops.append(waitcnt_lgkm(0))       # (the RGB layer's 1 / weight scale)
    for c in range(NC):
        ops += [ins for ins, _ in epilogue_ops(NT - 1, c)]
    return ops


def block_stream(opts):
    """f16p3s / mixs: the full path (the branch of the cut is an instruction of it, tagged 'cut': split_at_cut / skip_tail_ops)"""
    if EMB:      # ... and the block ends with the stores of its own tile's raw
        return [vm_wait(0), barrier()] + schedule(opts) + tail_ops() + raw_store_ops()
    return [waitcnt_vm(0), barrier()] + schedule(opts) + tail_ops()


def split_at_cut(body):
    """(instructions up to and including the cut's branch, the rest of the full path)"""
    i = next(k for k, ins in enumerate(body) if ins.tag == 'cut')
    return body[:i + 1], body[i + 1:]


# ---- f16c4e: the kernel around the block (one asm statement: setup, ring prologue, first tile's embedding, tile loop) --------------
KERNEL_OPERANDS = ('wimg', 'wave', 'ro', 'rd', 'z', 'raw', 'npts', 'S', 'zs', 'magic', 'sh1', 'sh2', 'tile', 'grid', 'ntiles')


def kernel_setup_ops():
    """(text lines, emulator function(st, args)) of the one-time setup: setup_ops' constants plus the operands of the kernel, the lane
    masks and the constants of the embedding"""
    L, emu0 = setup_ops()
    a = L.append
    for name, reg, wide in (('ro', S_PO, 1), ('rd', S_PD, 1), ('z', S_PZ, 1), ('raw', S_PRAW, 1), ('npts', S_NPTS, 0), ('S', S_S, 0), ('zs', S_ZS, 0),
                            ('magic', S_MAGIC, 0), ('sh1', S_SH1, 0), ('sh2', S_SH2, 0), ('tile', S_TILE, 0), ('grid', S_GRID, 0), ('ntiles', S_NTILES, 0)):
        a('s_mov_b%d %s, %%[%s]' % (64 if wide else 32, sreg(reg, 2 if wide else 1), name))
    a('s_sub_u32 %s, %s, 1' % (sreg(S_LAST), sreg(S_NPTS)))
    a('s_mov_b32 %s, 0x%08x' % (sreg(S_C25), f32_bits(0.25)))
    a('s_mov_b32 %s, 0x%08x' % (sreg(S_HI), f32_bits(INV2PI_HI)))
    a('s_mov_b32 %s, 0x%08x' % (sreg(S_LO), f32_bits(INV2PI_LO)))
    a('s_mov_b32 %s, 0x%08x' % (sreg(S_A16), f32_bits(ACT)))
    G0, G1 = V_G, V_G + 1
    a('v_lshrrev_b32_e32 %s, 4, %s' % (vreg(G0), vreg(V_LANE)))                # lane quarter q
    a('v_and_b32_e32 %s, 1, %s' % (vreg(G1), vreg(G0)))
    a('v_cmp_ne_u32_e64 %s, 0, %s' % (sreg(S_MQ1, 2), vreg(G1)))
    a('v_and_b32_e32 %s, 2, %s' % (vreg(G1), vreg(G0)))
    a('v_cmp_ne_u32_e64 %s, 0, %s' % (sreg(S_MQ2, 2), vreg(G1)))
    a('v_cmp_eq_u32_e64 %s, 3, %s' % (sreg(S_MQ3, 2), vreg(G0)))
    a('v_cmp_eq_u32_e64 %s, 0, %s' % (sreg(S_MQ0, 2), vreg(G0)))
    a('v_cmp_eq_u32_e64 %s, 1, %s' % (sreg(S_MQ1E, 2), vreg(G0)))
    a('v_cmp_gt_u32_e64 %s, 16, %s' % (sreg(S_ML16, 2), vreg(V_LANE)))
    a('v_and_b32_e32 %s, 15, %s' % (vreg(V_PL), vreg(V_LANE)))
    a('s_lshl_b32 %s, %s, 6' % (sreg(S_TB), sreg(S_WAVE)))
    a('v_add_u32_e32 %s, %s, %s' % (vreg(V_PL), sreg(S_TB), vreg(V_PL)))
    a('s_nop 3')                                                               # the masks are read by VALU from here on

    def emu(st, args):
        emu0(st)
        lanes = np.arange(64)
        q = lanes >> 4
        st.S[S_NPTS], st.S[S_LAST], st.S[S_S], st.S[S_ZS] = args['npts'], args['npts'] - 1, args['S'], args['zs']
        st.S[S_MAGIC], st.S[S_SH1], st.S[S_SH2] = args['magic'], args['sh1'], args['sh2']
        st.S[S_TILE], st.S[S_GRID], st.S[S_NTILES] = args['tile'], args['grid'], args['ntiles']
        st.S[S_C25], st.S[S_HI], st.S[S_LO], st.S[S_A16] = 0.25, float(INV2PI_HI), float(INV2PI_LO), ACT
        st.S[S_MQ1], st.S[S_MQ2], st.S[S_MQ3], st.S[S_MQ0], st.S[S_MQ1E] = (q & 1) != 0, (q & 2) != 0, q == 3, q == 0, q == 1
        st.S[S_ML16] = lanes < 16
        st.V[V_PL] = (st.wave * 64 + (lanes & 15)).astype(np.uint32)
    return L, emu


def div_magic(d):
    """(magic, sh1, sh2) with n / d = (t + ((n - t) >> sh1)) >> sh2, t = mulhi(n, magic), for every 32-bit n (Granlund & Montgomery,
    "Division by invariant integers using multiplication", fig. 4.1) -- restated by nerf_capi.hip for the kernel's operands"""
    assert 1 <= d < 2 ** 31
    l = (d - 1).bit_length()
    magic = (2 ** 32 * (2 ** l - d)) // d + 1
    return magic & 0xffffffff, min(l, 1), max(l - 1, 0)


def first_embed_ops():
    ops = []
    for c in range(NC):
        ops += emb_E_ops(c)
    for c in range(NC):
        ops += emb_V_ops(c)
    return ops + emb_V_commit_ops()


def kernel_text(opts):
    """the whole kernel as asm lines"""
    setup, _ = kernel_setup_ops()
    L = list(setup)
    a = L.append
    a('s_cmp_ge_u32 %s, %s' % (sreg(S_TILE), sreg(S_NTILES)))
    a('s_cbranch_scc1 L_exit_%=')
    L += [i.text for i in prologue_ops()]
    a('s_mov_b32 %s, %s' % (sreg(S_NEXT), sreg(S_TILE)))
    L += [i.text for i in raw_load_ops()]
    a('s_waitcnt vmcnt(0)')
    L += [i.text for i in first_embed_ops()]
    a('s_add_u32 %s, %s, %s' % (sreg(S_NEXT), sreg(S_TILE), sreg(S_GRID)))
    a('L_tile_%=:')
    body = block_stream(opts)
    L += [i.text for i in body]
    a('s_mov_b32 %s, %s' % (sreg(S_TILE), sreg(S_NEXT)))
    a('s_add_u32 %s, %s, %s' % (sreg(S_NEXT), sreg(S_NEXT), sreg(S_GRID)))
    a('s_cmp_lt_u32 %s, %s' % (sreg(S_TILE), sreg(S_NTILES)))
    a('s_cbranch_scc1 L_tile_%=')
    a('L_exit_%=:')
    a('s_waitcnt vmcnt(0)')
    a('s_mov_b32 m0, %s' % sreg(S_M0SAVE))
    return L, body


def emit_kernel(dirname, opts):
    lines, body = kernel_text(opts)
    n = {}
    for ins in body:
        n[ins.kind] = n.get(ins.kind, 0) + 1
    with open(os.path.join(dirname, 'nerf_mlp%s_asm.inc' % SUFFIX), 'w') as f:
        f.write('// GENERATED by gen/nerf_gen.py (NERF_GEN_FMT=%s) -- do not edit.  The fp16-only teacher chain as one statement: ring prologue, first '
                "tile's embedding, tile loop (ray loads, eleven layers, the next tile's embedding as fillers, raw stores).  Per tile: %s\n" %
                (FMT, ', '.join('%s %d' % kv for kv in sorted(n.items()))))
        for line in lines:
            f.write('"%s\\n\\t"\n' % line)
    regs = ['v%d' % i for i in range(256)] + ['a%d' % i for i in range(256)] + ['s%d' % i for i in range(N_SGPR_LO, N_SGPR_HI)] + ['vcc', 'scc', 'memory']
    with open(os.path.join(dirname, 'nerf_mlp%s_clobbers.inc' % SUFFIX), 'w') as f:
        f.write('// GENERATED by gen/nerf_gen.py: registers the kernel statement owns\n' + ', '.join('"%s"' % r for r in regs) + '\n')
    return n, body


def emulate_kernel(opts, img, aux, rays_o, rays_d, z, S, z_stride, n_pts, wave=0, block=0, grid=1, check_hazards=True, body=None):
    """one wave of workgroup `block` of a `grid`-workgroup launch over n_pts points (rays_o / rays_d [n_rays, 3], z [n_rays, S] or
    [S] with z_stride 0): returns (raw [n_pts, 4] float32 with the rows this wave stored filled in, NaN elsewhere; errors)"""
    body = body or block_stream(opts)
    st = NState(wave, img, aux)
    n_tiles = (n_pts + 64 * NC - 1) // (64 * NC)
    magic, sh1, sh2 = div_magic(S)
    _, setup = kernel_setup_ops()
    setup(st, dict(npts=n_pts, S=S, zs=z_stride, magic=magic, sh1=sh1, sh2=sh2, tile=block, grid=grid, ntiles=n_tiles))
    st.gmem = {'rays_o': np.ascontiguousarray(rays_o, dtype=np.float32).reshape(-1), 'rays_d': np.ascontiguousarray(rays_d, dtype=np.float32).reshape(-1),
               'z': np.ascontiguousarray(z, dtype=np.float32).reshape(-1), 'raw': np.full(n_pts * 4, np.nan, dtype=np.float32)}
    errs = []
    if block < n_tiles:
        st.run(prologue_ops())
        st.S[S_NEXT] = st.S[S_TILE]
        st.run(raw_load_ops())
        st.run([vm_wait(0)])
        st.run(first_embed_ops())
        st.S[S_NEXT] = st.S[S_TILE] + st.S[S_GRID]
        while True:
            st.run(body)
            st.S[S_TILE] = st.S[S_NEXT]
            st.S[S_NEXT] = st.S[S_NEXT] + st.S[S_GRID]
            if not st.S[S_TILE] < st.S[S_NTILES]:
                break
    errs += list(st.errors)
    if st.pend_ds:
        errs.append('%d LDS reads never waited for' % len(st.pend_ds))
    if check_hazards:
        errs += check_hazards_stream(body)
        errs += check_hazards_stream(first_embed_ops())
    return st.gmem['raw'].reshape(n_pts, 4), errs



def emit(dirname, opts):
    setup, _ = setup_ops()
    body = block_stream(opts)
    drop = getattr(opts, 'drop', ())          # diagnostics only (wrong results): timing knock-outs of instruction classes
    if drop:
        def keep(i):
            if i.kind == 'wait':
                return not (('lgkm' in drop and 'lgkmcnt' in i.text) or ('vm' in drop and 'vmcnt' in i.text))
            return i.kind not in drop
        body = [i for i in body if keep(i)]
    n = {}
    for ins in body:
        n[ins.kind] = n.get(ins.kind, 0) + 1
    with open(os.path.join(dirname, 'nerf_mlp%s_asm.inc' % SUFFIX), 'w') as f:
        f.write('// GENERATED by gen/nerf_gen.py -- do not edit.  One 128-point tile of the teacher MLP: %s\n' %
                ', '.join('%s %d' % kv for kv in sorted(n.items())))
        lines = setup + [i.text for i in body]
        if SKIPV:
            lines += ['s_branch L_done_%=', 'L_skip_%=:'] + [i.text for i in skip_tail_ops()] + ['L_done_%=:']
        for line in lines + ['s_mov_b32 m0, %s' % sreg(S_M0SAVE)]:
            f.write('"%s\\n\\t"\n' % line)
    with open(os.path.join(dirname, 'nerf_mlp%s_pro_asm.inc' % SUFFIX), 'w') as f:
        f.write('// GENERATED by gen/nerf_gen.py -- do not edit.  Ring prologue: chunks 0..2 of the stream\n')
        for line in setup + [i.text for i in prologue_ops()] + ['s_mov_b32 m0, %s' % sreg(S_M0SAVE)]:
            f.write('"%s\\n\\t"\n' % line)
    def clob(nv0, nv1, na):
        regs = ['v%d' % i for i in range(nv0, nv1)] + ['a%d' % i for i in range(na)]
        if na and A_EXTRA_CLOBBER:
            regs += ['a%d' % i for i in range(*A_EXTRA_CLOBBER)]
        regs += ['s%d' % i for i in range(N_SGPR_LO, N_SGPR_HI)] + ['vcc', 'scc', 'memory']
        return ', '.join('"%s"' % r for r in regs) + '\n'

    with open(os.path.join(dirname, 'nerf_mlp%s_clobbers.inc' % SUFFIX), 'w') as f:
        f.write('// GENERATED by gen/nerf_gen.py: registers the tile block owns\n' + clob(0, N_VGPR_CLOBBER, N_AGPR_CLOBBER))
    with open(os.path.join(dirname, 'nerf_mlp%s_pro_clobbers.inc' % SUFFIX), 'w') as f:
        f.write('// GENERATED by gen/nerf_gen.py: registers the ring prologue owns\n' + clob(V_L0, N_VGPR_CLOBBER, 0))
    return n, body


# ---------------------------------------------------------------------------------------------
# emulation of one wave over one tile (tests)
# ---------------------------------------------------------------------------------------------
def emulate_tile(opts, img, aux, frags, wave=0, n_tiles=1, check_hazards=True, body=None):
    """frags: {input name: uint32 [4, 64]} (INPUT_NAMES).  Returns (outputs float32 [8, 64], errors)."""
    body = body or block_stream(opts)
    st = NState(wave, img, aux)
    _, setup = setup_ops()
    setup(st)
    for name, reg in INPUT_NAMES:
        st.A[reg:reg + 4] = frags[name]
    st.run(prologue_ops())
    st.skipped = []
    for _ in range(n_tiles):
        st.out = {}
        if SKIPV:        # the emulated wave decides alone (the kernel ORs four waves): up to the branch, then one of the two exits
            pre, post = split_at_cut(body)
            st.run(pre)
            st.skipped.append(bool(st.skip))
            st.run(skip_tail_ops() if st.skip else post)
        else:
            st.run(body)
    errs = list(st.errors)
    if st.pend_ds:
        errs.append('%d LDS reads never waited for' % len(st.pend_ds))
    if check_hazards:
        errs += check_hazards_stream(body)
        if SKIPV:
            errs += check_hazards_stream(split_at_cut(body)[0] + skip_tail_ops())
    out = np.stack([st.out[k] for k in range(4 * NC)]).view(np.float32)
    if SKIPV:
        return out, errs, st.skipped
    return out, errs


def model_cycles(body):
    return B.model_cycles(body)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--emit', help='directory for nerf_mlp_asm.inc / nerf_mlp_pro_asm.inc')
    ap.add_argument('--dump', help='write the tile block as plain text')
    ap.add_argument('--lead', type=int)
    ap.add_argument('--lead6', type=int)
    ap.add_argument('--cap', type=int)
    ap.add_argument('--dma-gap', type=int)
    ap.add_argument('--wait-group', type=int)
    ap.add_argument('--pair', action='store_true', help='fillers only behind the second MFMA of a column-tile pair')
    ap.add_argument('--drop', default='', help='diagnostics only: comma list of instruction classes left out of the emitted '
                    'text (lgkm, dma, valu, ds, mfma6, mfma16, salu, nop): timing knock-outs, wrong results')
    a = ap.parse_args()
    given = {k: v for k, v in dict(lead=a.lead, lead6=a.lead6, cap=a.cap, dma_gap=a.dma_gap, wait_group=a.wait_group).items() if v is not None}
    opts = Opts(pair=a.pair, drop=tuple(x for x in a.drop.split(',') if x), **given)
    print('format', FMT, 'tiles', NT, 'chunks', NCH, 'MFMAs', N_ANCH, 'stream bytes', STREAM_BYTES)
    if a.emit:
        n, body = (emit_kernel if EMB else emit)(a.emit, opts)
        print('wrote', a.emit, n, 'model cycles per tile', model_cycles(body))
    if a.dump:
        body = block_stream(opts)
        with open(a.dump, 'w') as f:
            for ins in body:
                f.write(ins.text + '\n')
        print('model cycles per tile', model_cycles(body))


if __name__ == '__main__':
    sys.exit(main())
