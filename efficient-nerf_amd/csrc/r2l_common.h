// Shared host/device definitions for the R2L W256 ResMLP kernel: the packed weight
// image ("chunk stream") layout and the index maps both sides must agree on.
//
// Geometry (see DESIGN.md "K3"):
//   * one wave64 owns 32 rays; its activations never leave registers.  A layer is
//     Y^T[256 x 32] = W[256 x K] * X^T[K x 32] on v_mfma_f32_32x32x16_f16: W fragments are
//     the A operand (streamed through LDS, shared by the 4 waves of the workgroup), X^T
//     fragments the B operand (registers).  The 32x32 f32 result of output-feature tile t
//     has its ray on the lane and its 16 features in registers, which is exactly the B
//     fragment pair of k-steps 2t, 2t+1 of the next layer up to a fixed permutation of k
//     inside each 32-feature group; that permutation is applied to W on the host (kappa).
//   * the weight image is a sequence of equal-size chunks: FRAGS=16 fragments of 1 KiB
//     (64 lanes x 8 f16), NP parts each (hi [, lo]), plus a 1 KiB aux block (f32).
//       head  : 32 chunks; chunk c = k-steps 2c,2c+1 x 8 feature tiles (frag = ksl*8 + t)
//       body  : per layer 8 chunks; chunk t = feature tile t x 16 k-steps (frag = ks)
//       tail  : 1 chunk; rows 0..2 of tile 0 real, rest zero
//     aux (floats): body/tail [0..31] = bias*scale of the tile, [32] = 1/scale;
//                   head chunk 0 [0..255] = bias*scale, head chunk 31 [32] = 1/scale.
#pragma once
#include <stdint.h>

#define R2L_WIDTH 256
#define R2L_NTILE 8          // 256 / 32 output-feature tiles
#define R2L_KSTEPS 16        // 256 / 16 k-steps per body layer
#define R2L_NSAMPLE 16
#define R2L_NCOORD 48        // 16 samples x 3
#define R2L_L 10
#define R2L_EMBED 21         // 2L+1
#define R2L_IN 1008          // 48*21
#define R2L_HEAD_KSTEPS 64   // 63 real k-steps + 1 zero pad
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

R2L_HD int r2l_chunk_bytes(int np) { return (R2L_FRAGS * np) * R2L_FRAG_BYTES + R2L_AUX_BYTES; }
R2L_HD int r2l_chunks_per_tile(int n_block) { return R2L_HEAD_CHUNKS + 2 * n_block * R2L_NTILE + 1; }

// Input feature (0..255) that element j (0..7) of lane-half h (0..1) of body k-step ks
// (0..15) multiplies: the D-layout of v_mfma_f32_32x32x16 (row = (reg&3) + 8*(reg>>2) +
// 4*(lane>>5)) read back as a B fragment (reg = 8*(ks&1) + j of feature tile ks>>1).
R2L_HD int r2l_kappa(int ks, int h, int j) {
    return 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
}

// Column of head.0.weight (0..1007, or -1 = zero pad) that element j of lane-half h of
// head k-step ks (0..63) multiplies.  Reference embedding order per coordinate c is
// [sin(2^l x) l=0..9, cos(2^l x) l=0..9, x]  (model/nerf_raybased.py:198-208).
//   ks 0..47  : coordinate ks, frequency l=j; h=0 sin, h=1 cos
//   ks 48..59 : coordinate 4*(ks-48)+(j>>1), frequency 8+(j&1); h=0 sin, h=1 cos
//   ks 60..62 : identity of coordinate 16*(ks-60)+8h+j
//   ks 63     : pad
R2L_HD int r2l_head_col(int ks, int h, int j) {
    if (ks < 48) return ks * R2L_EMBED + (h ? R2L_L : 0) + j;
    if (ks < 60) return (4 * (ks - 48) + (j >> 1)) * R2L_EMBED + (h ? R2L_L : 0) + 8 + (j & 1);
    if (ks < 63) return (16 * (ks - 60) + 8 * h + j) * R2L_EMBED + 2 * R2L_L;
    return -1;
}
