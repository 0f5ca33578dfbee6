// Shared host/device definitions for the NeRF-teacher MLP kernel (8x256 + view branch,
// model/nerf_raybased.py:337-401): the fragment-stream layout of the packed weights.
//
// Same machine as the R2L kernel (r2l_common.h): one wave64 owns 32 POINTS, activations
// stay in registers as MFMA B fragments, weights stream global -> LDS ring -> A operand in
// chunks of 16 fragments (+ 1 KiB aux).  A layer with KS k-steps and NT output tiles is
// the fragment run [F0, F0 + NT*KS), tile-major / k inner; every layer starts on a chunk
// boundary.  Accumulators start from aux[32 * slot + ...] of the chunk that holds the
// tile's first fragment, slot = (first_frag % 16) / 2 (bias pre-multiplied by the layer's
// scale); the per-layer 1/scale travels as a kernel argument.
//
//   layer   input (K order)                          KS  NT  F0
//   L0      pts embedding E (63 + pad)                4   8     0   relu
//   L1..L4  256                                      16   8    32 + 128 i  relu
//   L5      256 (h) then E           [skip cat, :385] 20   8   544   relu
//   L6, L7  256                                      16   8   704, 832   relu
//   FA      256 -> feature(256) | alpha(1)           16   9   960   (no activation)
//   V       feature(256) then view embedding(27+pad) 18   4  1104   relu
//   RGB     128 -> 3                                  8   1  1176
#pragma once
#include "r2l_common.h"

#define NERF_FRAGS_TOTAL 1184
#define NERF_CHUNKS (NERF_FRAGS_TOTAL / R2L_FRAGS)  // 74
#define NERF_F0_L0 0
#define NERF_F0_L1 32
#define NERF_F0_L5 544
#define NERF_F0_L6 704
#define NERF_F0_FA 960
#define NERF_F0_V 1104
#define NERF_F0_RGB 1176
#define NERF_N_SCALES 12  // L0..L7, FA, V, RGB (+1 spare)
#define NERF_PTS_PER_WAVE 32
#define NERF_TILE_PTS 128

R2L_HD int nerf_aux_slot(int first_frag) { return (first_frag % R2L_FRAGS) / 2; }

// Reference column (0..62) of the pts embedding [x, sin(2^0 x), cos(2^0 x), ...]
// (utils/run_nerf_raybased_helpers.py:34-56) that element j of lane-half h of E k-step e
// holds; -1 = pad.
//   e = 0..2 : coordinate e, frequency j (0..7); h = 0 sin, 1 cos
//   e = 3    : j = 0..5 frequency 8 + (j&1) of coordinate j>>1 (h: sin|cos);
//              h=0: j=6 -> x0, j=7 -> x1;  h=1: j=6 -> x2, j=7 -> pad
R2L_HD int nerf_pts_col(int e, int h, int j) {
    if (e < 3) return 3 + j * 6 + (h ? 3 : 0) + e;
    if (j < 6) return 3 + (8 + (j & 1)) * 6 + (h ? 3 : 0) + (j >> 1);
    if (h == 0) return j - 6;          // x0, x1
    return j == 6 ? 2 : -1;            // x2, pad
}

// Reference column (0..26) of the view-direction embedding (multires_views = 4) held by
// element j of lane-half h of view k-step v (0..1); -1 = pad.
//   v = 0 : j = 0..3 frequency j of d0, j = 4..7 frequency j-4 of d1 (h: sin|cos)
//   v = 1 : j = 0..3 frequency j of d2; h=0: j=4..6 -> d0,d1,d2, j=7 pad; h=1: j>=4 pad
R2L_HD int nerf_view_col(int v, int h, int j) {
    if (v == 0) return 3 + (j & 3) * 6 + (h ? 3 : 0) + (j >> 2);
    if (j < 4) return 3 + j * 6 + (h ? 3 : 0) + 2;
    if (h == 0 && j < 7) return j - 4;
    return -1;
}
