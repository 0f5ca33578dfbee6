// Shared host/device definitions for the R2L W256 ResMLP kernel: the packed weight
// image ("chunk stream") layout and the index maps both sides must agree on.
//
// Geometry (see DESIGN.md "K3"):
//   * one wave64 owns 32 rays = 2 column tiles of 16; its activations never leave
//     registers.  A layer is Y^T[256 x 32] = W[256 x K] * X^T[K x 32] on
//     v_mfma_f32_16x16x32_f16: W fragments (16 output features x 32 inputs, 1 KiB) are the A
//     operand, streamed through LDS and shared by the 4 waves of the workgroup; X^T
//     fragments the B operand (registers).  The 16x16 f32 result of row tile u has its ray
//     on lane&15 and features 16u + 4*(lane>>4) + reg in its 4 registers; two consecutive
//     row tiles (2s, 2s+1) are exactly the B fragment of k-step s of the next layer up to a
//     fixed permutation of k inside each 32-feature group, applied to W on the host (kappa).
//   * the weight image is a sequence of equal-size chunks: FRAGS=16 fragments of 1 KiB
//     (64 lanes x 8 f16), NP parts each (hi [, lo]), plus a 1 KiB aux block (f32).
//       head  : 32 chunks; chunk c = k-step c (32 inputs) x 16 row tiles (frag = u)
//       body  : per layer 8 chunks; chunk m = row tiles 2m, 2m+1 x 8 k-steps (frag = (u&1)*8 + s)
//       tail  : 1 chunk; rows 0..2 of row tile 0 real (frags 0..7), rest zero
//     aux (floats): body/tail [0..31] = bias*scale of the chunk's 32 features, [32] = 1/scale;
//                   head chunk 0 [0..255] = bias*scale, head chunk 31 [32] = 1/scale.
#pragma once
#include <stdint.h>

#define R2L_WIDTH 256
#define R2L_RTILES 16        // 256 / 16 output-feature (row) tiles
#define R2L_KSTEPS 8         // 256 / 32 k-steps per body layer
#define R2L_CTILES 2         // 16-ray column tiles per wave
#define R2L_NSAMPLE 16
#define R2L_NCOORD 48        // 16 samples x 3
#define R2L_L 10
#define R2L_EMBED 21         // 2L+1
#define R2L_IN 1008          // 48*21
#define R2L_HEAD_KSTEPS 32   // 1008 -> 1024 inputs / 32
#define R2L_HEAD_CHUNKS 32
#define R2L_FRAGS 16
#define R2L_FRAG_BYTES 1024
#define R2L_AUX_BYTES 1024
#define R2L_RAYS_PER_WAVE 32
#define R2L_WAVES 4
#define R2L_TILE_RAYS (R2L_RAYS_PER_WAVE * R2L_WAVES)

#ifdef __HIPCC__
#define R2L_HD __host__ __device__ inline
#else
#define R2L_HD inline
#endif

// FP16_FP8 mode, body chunks only (head and tail keep the hi|lo fp16 layout): piece 2f = the fp16 hi
// fragment f = (u&1)*8 + s as above; piece 2f+1 = 16 bytes/lane of e4m3 operand of
// v_mfma_scale_f32_16x16x128_f8f6f4 for row tile u: s>>2 = term (0: (w - hi(w)) * 2^7 against the e5m2
// activations, 1: w * 2^-5 against the e5m2 activation residuals), (s>>1)&1 = K-step t of 128 inputs,
// s&1 = which 16 of the lane's 32 bytes.  Element j (0..31) of lane quarter q of K-step t multiplies
// input feature r2l_mix_feat(t, q, j): the 4 accumulator registers of row tile 8t + (j>>2).
#define R2L_MIX_WL_SHIFT 7    // (w*S - hi) * 2^7  : |.| <= 256
#define R2L_MIX_W_SHIFT 5     // (w*S) * 2^-5      : |.| <  256
R2L_HD int r2l_mix_feat(int t, int q, int j) { return 16 * (8 * t + (j >> 2)) + 4 * q + (j & 3); }

R2L_HD int r2l_chunk_bytes(int np) { return (R2L_FRAGS * np) * R2L_FRAG_BYTES + R2L_AUX_BYTES; }
R2L_HD int r2l_chunks_per_tile(int n_block) { return R2L_HEAD_CHUNKS + 2 * n_block * (R2L_RTILES / 2) + 1; }

// Input feature (0..255) that element j (0..7) of lane quarter q (= lane>>4, 0..3) of body
// k-step s (0..7) multiplies: the D layout of v_mfma_f32_16x16x32 (row = 4*(lane>>4) + reg)
// read back as a B fragment: elements 0..3 = registers of row tile 2s, 4..7 = of row tile 2s+1.
R2L_HD int r2l_kappa(int s, int q, int j) { return 32 * s + 16 * (j >> 2) + 4 * q + (j & 3); }

// The hand-scheduled body (r2l_body_kernel, FP16_FP8) uses the 32x32 MFMA shapes: a wave's 32 rays are ONE column tile,
// lane = 32 h + ray; accumulator register r of row tile u (32 features) = feature 32u + 8(r/4) + 4h + r%4.
// r2l_kappa32(s, h, j): input feature that element j (0..7) of lane half h of fp16 k-step s (0..15, 16 features each)
// multiplies: registers 4g .. 4g+3 of row tile u go to k-step 2u + (g>>1), elements 4(g&1) .. 4(g&1)+3.
// r2l_mix32(t, h, e): element e (0..31) of lane half h of K=64 step t = the fp16 k-steps 4t .. 4t+3 in order.
R2L_HD int r2l_kappa32(int s, int h, int j) { return 32 * (s >> 1) + 16 * (s & 1) + 8 * (j >> 2) + 4 * h + (j & 3); }
R2L_HD int r2l_mix32(int t, int h, int e) { return r2l_kappa32(4 * t + (e >> 3), h, e & 7); }
// The generated head layer (r2l_head_kernel, FP16_FP8; csrc/gen/head_gen.py): K = 64 group p = point p of the ray;
// r2l_head_col32(p, s, h, j): column of head.0.weight (or -1 = pad) that element j of lane half h of fp16 k-step s (0..3,
// 16 features each) of the group multiplies:
//   s < 3 : coordinate s, frequency j, sin (h = 0) | cos (h = 1)
//   s = 3 : j < 6: coordinate j >> 1, frequency 8 + (j & 1), sin | cos;  j = 6: x0 | x2;  j = 7: x1 | pad
// Stream of a tile: 32 chunks of 28 KiB = (point p, row tiles 4m .. 4m+3): piece 4k + s = fp16 fragment of k-step s of
// the chunk's k-th row tile (weights x act_scale), piece 16 + 2k + t = first 16 B/lane of bf6 operand t (0: (w - hi(w)) /
// 2^(e-16), 1: w / 2^(e-4)), piece 24 + k = last 8 B/lane of both (t * 512); then 2 KiB: 256 f32 bias x act_scale | at
// 1024: 4 x (swl, sw, 0, 0) E8M0.
R2L_HD int r2l_head_col32(int p, int s, int h, int j) {
    if (s < 3) return (3 * p + s) * R2L_EMBED + (h ? R2L_L : 0) + j;
    if (j < 6) return (3 * p + (j >> 1)) * R2L_EMBED + (h ? R2L_L : 0) + 8 + (j & 1);
    if (j == 6) return (3 * p + (h ? 2 : 0)) * R2L_EMBED + 2 * R2L_L;
    return h ? -1 : (3 * p + 1) * R2L_EMBED + 2 * R2L_L;
}
// Body stream of R2L_PREC_FP16_FP8 (r2l_capi.hip pack_body_v3, gen/body_gen.py configure): 28 KiB chunks, every operand of the
// correction terms streamed ('bf6').  -DR2L_BF6R_STREAM builds the round-3 experiment 'bf6r' (22 KiB chunks: the bf6(W) operands
// are converted from the fp16 fragments in registers; generate the two r2l_body*_asm.inc with `body_gen.py --fmt bf6r`): parity
// green, 5 % slower (the 64 conversions per block are 32 VALU cycles each and do not hide: profiles/r03_dma_experiments.txt).
#ifdef R2L_BF6R_STREAM
#define R2L_BF6_CHUNK 22528
#else
#define R2L_BF6_CHUNK 28672
#endif
#define R2L_HEAD_STREAM_BYTES (32 * 28672)
#define R2L_HEAD_AUX_BYTES 2048
#define R2L_HEAD_LDS (4 * 28672 + R2L_HEAD_AUX_BYTES)
// R2L_PREC_FP16X3_ASM head stream (gen/head_gen.py --fmt f16): 32 chunks of 32 KiB = 16 hi(W) fragments + 16 fragments
// fp16(w - hi(w)) at pieces 16 + 4 k + s; same aux block (the scale words are unused)
#define R2L_HEADX_CHUNK 32768
#define R2L_HEADX_STREAM_BYTES (32 * R2L_HEADX_CHUNK)
#define R2L_HEADX_LDS (4 * R2L_HEADX_CHUNK + R2L_HEAD_AUX_BYTES)
// aux block of the body stream (4 KiB per ResMLP block): 256 f32 bias | at 1024: 4 x (swl1, sw1, swl2, sw2) E8M0 |
// at 1088: the biased activation exponents (127 + E_in, 127 + E_h, 127 + E_out, 0) as dwords, twice
#define R2L_BODY_AUX_BYTES 4096
#define R2L_BODY_AUX_ACT 1088
#define R2L_ACT_EXP 3          // default: activations (act_scale domain) / 2^3 fit bf6
#define R2L_ACT_EXP_MIN (-8)
#define R2L_ACT_EXP_MAX 14
// register image of x in HBM (head -> body -> tail): [tile][wave][group 0..31][lane 0..63][4] f32, group = 4u + g
R2L_HD int r2l_x_group(int feature) { return 4 * (feature >> 5) + ((feature >> 3) & 3); }   // of features f .. f+3, f % 4 == 0
R2L_HD int r2l_x_half(int feature) { return (feature >> 2) & 1; }

// Column of head.0.weight (0..1007, or -1 = zero pad) that element j of lane quarter q of
// head k-step s (0..31) multiplies.  Reference embedding order per coordinate c is
// [sin(2^l x) l=0..9, cos(2^l x) l=0..9, x]  (model/nerf_raybased.py:198-208).
//   s 0..23  : coordinate 2s + (q>>1), frequency l=j; q&1 = 0 sin, 1 cos
//   s 24..29 : coordinate 8(s-24) + 2q + (j>>2), frequency 8 + (j&1); (j>>1)&1 = 0 sin, 1 cos
//   s 30     : identity of coordinate 8q + j
//   s 31     : identity of coordinate 32 + 8q + j for q < 2, pad for q >= 2
R2L_HD int r2l_head_col(int s, int q, int j) {
    if (s < 24) return (2 * s + (q >> 1)) * R2L_EMBED + ((q & 1) ? R2L_L : 0) + j;
    if (s < 30) return (8 * (s - 24) + 2 * q + (j >> 2)) * R2L_EMBED + (((j >> 1) & 1) ? R2L_L : 0) + 8 + (j & 1);
    if (s == 30) return (8 * q + j) * R2L_EMBED + 2 * R2L_L;
    if (q < 2) return (32 + 8 * q + j) * R2L_EMBED + 2 * R2L_L;
    return -1;
}
