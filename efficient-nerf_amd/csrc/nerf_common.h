// Shared host/device definitions for the NeRF-teacher MLP kernel (8x256 + view branch,
// model/nerf_raybased.py:337-401): the fragment-stream layout of the packed weights.
//
// Same machine as the R2L kernel (r2l_common.h): one wave64 owns 32 POINTS (2 column tiles
// of 16), activations stay in registers as v_mfma_f32_16x16x32_f16 B fragments, weights
// stream global -> LDS ring -> A operand in chunks of 16 fragments (+ 1 KiB aux).  A layer
// with KS k-steps (32 inputs each) and RT row tiles (16 outputs each) is the fragment run
// [F0, F0 + RT*KS), row-tile-major / k inner.  Accumulators start from aux[16 * slot + ...] of
// the chunk that holds the row tile's first fragment, slot = (first_frag % 16) / 2 (bias
// pre-multiplied by the layer's scale); the per-layer 1/scale travels as a kernel argument.
//
//   layer   input (K order)                          KS  RT  F0
//   L0      pts embedding E (63 + pad)                2  16     0   relu
//   L1..L4  256                                       8  16    32 + 128 i  relu
//   L5      256 (h) then E           [skip cat, :385] 10  16   544   relu
//   L6, L7  256                                       8  16   704, 832   relu
//   FA      256 -> feature(256) | alpha(1)            8  17   960   (no activation)
//   V       feature(256) then view embedding(27+pad)  9   8  1096   relu
//   RGB     128 -> 3                                  4   1  1168   (+12 unused fragments)
#pragma once
#include "r2l_common.h"

#define NERF_FRAGS_TOTAL 1184
#define NERF_CHUNKS (NERF_FRAGS_TOTAL / R2L_FRAGS)  // 74
#define NERF_F0_L0 0
#define NERF_F0_L1 32
#define NERF_F0_L5 544
#define NERF_F0_L6 704
#define NERF_F0_FA 960
#define NERF_F0_V 1096
#define NERF_F0_RGB 1168
// FP16_FP8: stream of gen/nerf_gen.py (80 chunks of whole row tiles, 1 KiB pieces) followed by the resident table
#define NERF_CHAIN_STREAM_BYTES 2166784
// FP16X1: the same chain without its correction terms (NERF_GEN_FMT=f16: no bf6 operand pieces, 44 chunks of up to 4 row tiles, same table)
#define NERF_CHAINX_STREAM_BYTES 1298432
// FP16X3_ASM: fp16x3's arithmetic on the chain (NERF_GEN_FMT=f16p3: a hi and a lo fragment of W x 2^k per k-step, 84 chunks, same table
// with 2^-k at the scale bytes' place)
#define NERF_CHAINP3_STREAM_BYTES 2433024
// R2L_PREC_FP16_MIX (gen/nerf_gen.py NERF_GEN_FMT=mix): the bf6 chain with trunk layers L1 .. L<NERF_MIX_K> in three fp16 passes (their
// chunks in the hi | lo layout of the p3 stream, W x 2^k): 80 chunks as the bf6 stream, 2 x 8 of them 32 KiB instead of 28
// the three-pass chain without its view branch (gen/nerf_gen.py NERF_GEN_FMT=f16p3a): trunk + the alpha row (+ one all-zero row tile), 68 chunks
#define NERF_CHAINP3A_STREAM_BYTES 1998848
// the chains with a second exit behind the density (gen/nerf_gen.py NERF_GEN_FMT=f16p3s / mixs): the feature | alpha layer split into layer A
// (the alpha row, one chunk) and layer F (the feature rows, 8 chunks): 12 layers, 16 KiB less stream than the unsplit chains
#define NERF_CHAINP3S_STREAM_BYTES 2416640
#define NERF_CHAINMS_STREAM_BYTES 2220032
#define NERF_MIX_K 2
#define NERF_CHAINM_STREAM_BYTES 2232320
#define NERF_CHAIN_AUX_BYTES 16384
#define NERF_CHAIN_AUX_LAYER 1280   // per layer: 272 f32 bias | at byte 1152: 4 lane quarters x (swl, sw, 0, 0)
#define NERF_CHAIN_AUX_SCALES 1152
#define NERF_CHAIN_RING_BYTES (4 * 32768)
#define NERF_CHAIN_LDS (NERF_CHAIN_RING_BYTES + NERF_CHAIN_AUX_BYTES)
#define NERF_CHAIN_LDS_SKIP (NERF_CHAIN_LDS + 16)   // + the two LDS words the second exit's OR alternates between (at byte NERF_CHAIN_LDS)
#define NERF_N_SCALES 12  // L0..L7, FA, V, RGB (+1 spare)
#define NERF_PTS_PER_WAVE 32
#define NERF_TILE_PTS 128

R2L_HD int nerf_aux_slot(int first_frag) { return (first_frag % R2L_FRAGS) / 2; }

// Reference column (0..62) of the pts embedding [x, sin(2^0 x), cos(2^0 x), ...]
// (utils/run_nerf_raybased_helpers.py:34-56) that element j of lane quarter q of E k-step e
// (0..1) holds; -1 = pad.  Column of (coordinate k, frequency l): 3 + 6l + (cos ? 3 : 0) + k.
//   e = 0 : coordinate q>>1, frequency j; q&1 = 0 sin, 1 cos
//   e = 1 : q < 2: coordinate 2, frequency j, q&1 sin|cos
//           q = 2: j<6 sin of frequency 8+(j&1) of coordinate j>>1; j=6 -> x0, j=7 -> x1
//           q = 3: j<6 cos of the same;                              j=6 -> x2, j=7 -> pad
R2L_HD int nerf_pts_col(int e, int q, int j) {
    if (e == 0) return 3 + j * 6 + ((q & 1) ? 3 : 0) + (q >> 1);
    if (q < 2) return 3 + j * 6 + ((q & 1) ? 3 : 0) + 2;
    if (j < 6) return 3 + (8 + (j & 1)) * 6 + ((q == 3) ? 3 : 0) + (j >> 1);
    if (q == 2) return j - 6;          // x0, x1
    return j == 6 ? 2 : -1;            // x2, pad
}

// Reference column (0..26) of the view-direction embedding (multires_views = 4) held by
// element j of lane quarter q of the single view k-step; -1 = pad.
//   q < 3 : direction component q: j&3 = frequency, j>>2 = 0 sin, 1 cos
//   q = 3 : j < 3 -> identity of component j, else pad
R2L_HD int nerf_view_col(int q, int j) {
    if (q < 3) return 3 + (j & 3) * 6 + ((j >> 2) ? 3 : 0) + q;
    return j < 3 ? j : -1;
}
