// NeRF teacher hot path for MI355X / gfx950, hand-written HIP.
//
//   nerf_mlp_kernel<NP>     K4+K5 fused: point = o + d*z, Embedder(pts, L=10) + Embedder(viewdir,
//                           L=4), the 8x256 MLP with the skip concat, feature/alpha heads, the
//                           view branch and the rgb head -> raw[pt] = (rgb, sigma).  Same machine
//                           as r2l_resmlp_kernel: a wave owns 32 points, activations stay in
//                           registers as MFMA B fragments, weights stream through the LDS ring.
//   nerf_raw2outputs_kernel K6: alpha compositing, one wave per ray, prefix product in a wave scan
//   nerf_sample_pdf_kernel  K7: inverse-CDF sampling, wave prefix sum + per-lane binary search in LDS
//   nerf_merge_kernel       K8: merge of two sorted rows by rank (merge path), no sort
//   nerf_get_rays_kernel    get_rays
//
// Reference semantics (file:line in MingSun-Tse/Efficient-NeRF):
//   get_rays        utils/run_nerf_raybased_helpers.py:231-257
//   viewdirs        main.py:148-157
//   pts             main.py:701-702 / 733-734
//   Embedder        utils/run_nerf_raybased_helpers.py:24-56
//   NeRF.forward    model/nerf_raybased.py:377-401
//   raw2outputs     main.py:556-621
//   sample_pdf      utils/run_nerf_raybased_helpers.py:283-330 (det=True)
//   sort-merge      main.py:730-732
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/r2l_hip.h"
#include "nerf_kernels.h"
#include "r2l_device.h"

// ====================================================================================
// MLP
// ====================================================================================
enum { EPI_RELU = 0, EPI_LINEAR_ALPHA = 1, EPI_RGB = 2 };

struct MlpOut {       // valid in the quarter-0 lanes (q == 0), one value per column tile
    float alpha[2];   // sigma (alpha_linear output)
    float rgb[2][3];  // rgb_linear output
};

// One activation set (256 features x 32 points per wave) as MFMA B operands: fp16 hi and lo fragments
struct ActSet {
    f16x8 h[8][2], l[8][2];
};

// epilogue of accumulator registers 2*pair, 2*pair+1 of row tile u, column tile c, of a layer with RT row tiles
template <int NP, int EPI, int RT>
__device__ __forceinline__ void mlp_epi_pair(const f32x4& acc, float inv, ActSet& D, int u, int c, int pair,
                                             float act_scale, float neg1, MlpOut& out) {
    if (EPI == EPI_RGB) {
        if (pair == 0) {
            out.rgb[c][0] = acc[0] * inv;
            out.rgb[c][1] = acc[1] * inv;
        } else {
            out.rgb[c][2] = acc[2] * inv;
        }
        return;
    }
    if (EPI == EPI_LINEAR_ALPHA && u == RT - 1) {  // row tile 16 of FA: row 0 = alpha_linear
        if (pair == 0) out.alpha[c] = acc[0] * inv;
        return;
    }
    float v[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        v[k] = acc[2 * pair + k] * inv;
        if (EPI == EPI_RELU) v[k] = fmaxf(v[k], 0.0f);
        v[k] *= act_scale;
    }
    const int idx = 2 * (u & 1) + pair;
    split_store2<NP>(v[0], v[1], D.h[(u >> 1) & 7][c], D.l[(u >> 1) & 7][c], idx, neg1);
}

// One Linear layer = fragment run [F0, F0 + RT*KS).  Input fragments: k-steps 0..7 from the set S,
// k-steps 8.. from (Xh,Xl) (embedding fragments).  Output row tiles are written as fragments of D.
// The epilogue of row tile u-1 is interleaved with the MFMAs of row tile u; the last one's is exposed.
template <int NP, int KS, int RT, int F0, int EPI>
__device__ __forceinline__ void mlp_layer(Ring<NP>& R, const ActSet& S, const f16x8 (&Xh)[2][2],
                                          const f16x8 (&Xl)[2][2], ActSet& D, float inv, float act_scale, float neg1,
                                          int q, MlpOut& out) {
    f32x4 acc[2], prev[2];
#pragma unroll
    for (int u = 0; u < RT; ++u) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int fq = F0 + u * KS + s;
            const int pos = fq % R2L_FRAGS;
            ring_step<NP>(R, pos);  // pos == 8: rendezvous + refill
            if (s == 0) {
                acc[0] = acc_init<NP>(R.use_off, 16 * nerf_aux_slot(fq), q);
                acc[1] = acc[0];
            }
            AFrag<NP> nxt = (pos + 1 < R2L_FRAGS) ? read_frag<NP>(R.use_off + R.lane * 16, pos + 1)
                                                  : read_frag<NP>(ring_next_off<NP>(R.use_off) + R.lane * 16, 0);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (s < 8) acc[c] = mfma_step<NP>(R.pre, S.h[s < 8 ? s : 0][c], S.l[s < 8 ? s : 0][c], acc[c]);
                else acc[c] = mfma_step<NP>(R.pre, Xh[s >= 8 ? s - 8 : 0][c], Xl[s >= 8 ? s - 8 : 0][c], acc[c]);
            }
            R.pre = nxt;
            if (u > 0) {
#pragma unroll
                for (int i = (4 * s + KS - 1) / KS; i < (4 * (s + 1) + KS - 1) / KS && i < 4; ++i)
                    mlp_epi_pair<NP, EPI, RT>(prev[i >> 1], inv, D, u - 1, i >> 1, i & 1, act_scale, neg1, out);
            }
            if (pos == R2L_FRAGS - 1) ring_next<NP>(R);
        }
        prev[0] = acc[0];
        prev[1] = acc[1];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) mlp_epi_pair<NP, EPI, RT>(prev[i >> 1], inv, D, RT - 1, i >> 1, i & 1, act_scale, neg1, out);
}

// Inputs of one 128-point tile for this lane (a lane serves point (lane & 15) of both column tiles of its wave), in two
// steps so that the chain kernel can fetch the next tile's values before it enters its layer block:
//   nerf_tile_load   ray origin / direction / depth (and the given view direction) of the lane's two points
//   nerf_tile_embed  pts = rays_o + rays_d * z (main.py:701), view directions (main.py:148-162) and the fragments of both
//                    embeddings (nerf_common.h: nerf_pts_col / nerf_view_col), act_scale folded in, fp16 hi | lo B operands
template <int NC, bool VD = true>
struct NerfTileRawT {       // NC column tiles of 16 points per wave (2: 128-point workgroup tiles; 3 / 4: 192 / 256, the fp16x1 chain)
    float o[NC][3], d[NC][3], v[VD ? NC : 1][3], z[NC];      // VD = false: no given view directions (they are rays_d / |rays_d|)
};
typedef NerfTileRawT<2> NerfTileRaw;

// n_pts < 2^31 (checked by the host): 32-bit point and ray indices
template <int NC, bool VD>
__device__ __forceinline__ void nerf_tile_load(const NerfMlpParams& p, int tile, int wave, int lane, NerfTileRawT<NC, VD>& r) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const unsigned pt_raw = (unsigned)tile * (64 * NC) + wave * (16 * NC) + c * 16 + (lane & 15);
        const unsigned last = (unsigned)p.n_pts - 1u;
        const unsigned pt = pt_raw < last ? pt_raw : last;   // tail lanes repeat the last point (their stores are masked)
        const unsigned ray = pt / (unsigned)p.S;
        const unsigned smp = pt - ray * (unsigned)p.S;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            r.o[c][k] = p.rays_o[(size_t)ray * 3 + k];
            r.d[c][k] = p.rays_d[(size_t)ray * 3 + k];
            if constexpr (VD) r.v[c][k] = p.viewdirs ? p.viewdirs[(size_t)ray * 3 + k] : 0.0f;
        }
        r.z[c] = p.z[(size_t)ray * p.z_stride + smp];
    }
}

template <int NP, int NC = 2, bool VD = true>
__device__ __forceinline__ void nerf_tile_embed(const NerfMlpParams& p, const NerfTileRawT<NC, VD>& r, int lane, f16x8 (&Eh)[2][NC],
                                                f16x8 (&El)[2][NC], f16x8 (&Vh)[2][NC], f16x8 (&Vl)[2][NC]) {
    const int q = lane >> 4;
    const float act_scale = p.act_scale;
    const bool is_cos = (q & 1) != 0;   // odd lane quarters hold cosines in every k-step (nerf_common.h)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float* d = r.d[c];
        // viewdirs = rays_d / ||rays_d||  (main.py:154-156); NDC renders carry those of the world-space rays (:148-162)
        const float nrm = sqrtf(__fadd_rn(__fadd_rn(d[0] * d[0], d[1] * d[1]), d[2] * d[2]));
        float xs[3], vs[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            xs[k] = __fadd_rn(r.o[c][k], __fmul_rn(d[k], r.z[c]));  // rays_o + rays_d * z  (main.py:701)
            if constexpr (VD) vs[k] = p.viewdirs ? r.v[c][k] : __fdiv_rn(d[k], nrm);
            else vs[k] = __fdiv_rn(d[k], nrm);
        }
        const Rev r0 = to_rev(xs[0]), r1 = to_rev(xs[1]), r2 = to_rev(xs[2]);
        {   // E step 0: coordinate q>>1, frequencies 0..7, sin|cos by q&1
            const Rev rr = (q & 2) ? r1 : r0;
            float pw = 1.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                split_store<NP>(trig_pow2(rr, pw, is_cos) * act_scale, Eh[0][c], El[0][c], j);
                pw *= 2.0f;
            }
        }
        {   // E step 1: q<2: coordinate 2, frequencies 0..7; q>=2: frequencies 8,9 of all three, then the identity.
            // One evaluation per element: the lane quarter selects the argument, not the result.
            float pw = 1.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const Rev rh = j < 2 ? r0 : (j < 4 ? r1 : r2);                 // q >= 2, j < 6
                Rev rr;
                rr.rh = (q & 2) ? rh.rh : r2.rh;
                rr.rl = (q & 2) ? rh.rl : r2.rl;
                const float pq = (q & 2) ? ((j & 1) ? 512.0f : 256.0f) : pw;
                float val = trig_pow2(rr, pq, is_cos);
                if (j == 6) val = (q & 2) ? (q == 3 ? xs[2] : xs[0]) : val;
                if (j == 7) val = (q & 2) ? (q == 3 ? 0.0f : xs[1]) : val;
                split_store<NP>(val * act_scale, Eh[1][c], El[1][c], j);
                pw *= 2.0f;
            }
        }
        {   // view step: q<3: component q, frequencies j&3, sin|cos by j>>2; q=3: identity
            const Rev rv = to_rev(q == 0 ? vs[0] : (q == 1 ? vs[1] : vs[2]));
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float t = trig_pow2(rv, (float)(1 << (j & 3)), (j >> 2) != 0);
                const float idv = j < 3 ? vs[j < 3 ? j : 0] : 0.0f;
                split_store<NP>(((q == 3) ? idv : t) * act_scale, Vh[0][c], Vl[0][c], j);
            }
        }
    }
}

template <int NP>
__global__ __launch_bounds__(256, 1) void nerf_mlp_kernel(NerfMlpParams p) {
    typedef KCfg<NP> C;
    Ring<NP> R;
    R.wimg = p.wimg;
    R.cpt = NERF_CHUNKS;
    R.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    R.lane = threadIdx.x & 63;
    R.issue_pos = 0;
    R.issue_off = 0;
    R.use_off = 0;
    const int lane = R.lane;
    const int q = lane >> 4;
    const float act_scale = p.act_scale;
    const float neg1 = p.neg1;

#pragma unroll
    for (int i = 0; i < C::D; ++i) ring_issue<NP>(R);
    R2L_WAIT_VMCNT(C::WAIT_PRO);
    __builtin_amdgcn_s_barrier();
    R.pre = read_frag<NP>(lane * 16, 0);

    ActSet A1, A2;
    f16x8 Eh[2][2], El[2][2], Vh[2][2], Vl[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {  // unused second slot of the "extra" operand of the V layer
        Vh[1][c] = (f16x8)(f16)0;
        Vl[1][c] = (f16x8)(f16)0;
    }

    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        NerfTileRaw traw;
        nerf_tile_load(p, tile, R.wave, lane, traw);
        nerf_tile_embed<NP>(p, traw, lane, Eh, El, Vh, Vl);

        MlpOut out;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            out.alpha[c] = 0.f;
            out.rgb[c][0] = out.rgb[c][1] = out.rgb[c][2] = 0.f;
        }
        // L0: E -> A2   (KS = 2: the source operand is E parked in A1[0..1], always hi/lo fp16)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                A1.h[e][c] = Eh[e][c];
                A1.l[e][c] = El[e][c];
            }
        mlp_layer<NP, 2, 16, NERF_F0_L0, EPI_RELU>(R, A1, Eh, El, A2, p.inv_scale[0], act_scale, neg1, q, out);
        for (int it = 0; it < 3; ++it) {
            if (it == 2) {
                // L5: [h(256) | E] -> A1, then move to A2 so the two-layer body is reused
                mlp_layer<NP, 10, 16, NERF_F0_L5, EPI_RELU>(R, A2, Eh, El, A1, p.inv_scale[5], act_scale, neg1,
                                                                    q, out);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        A2.h[i][c] = A1.h[i][c];
                        A2.l[i][c] = A1.l[i][c];
                    }
                }
            }
            // (L1,L2) (L3,L4) (L6,L7): the fragment run of the pair is contiguous per iteration
            const float inva = p.inv_scale[it == 0 ? 1 : (it == 1 ? 3 : 6)];
            const float invb = p.inv_scale[it == 0 ? 2 : (it == 1 ? 4 : 7)];
            mlp_layer<NP, 8, 16, NERF_F0_L1, EPI_RELU>(R, A2, Eh, El, A1, inva, act_scale, neg1, q, out);
            mlp_layer<NP, 8, 16, NERF_F0_L1 + 128, EPI_RELU>(R, A1, Eh, El, A2, invb, act_scale, neg1, q, out);
        }
        // FA: feature_linear | alpha_linear (no activation) -> A1, sigma
        mlp_layer<NP, 8, 17, NERF_F0_FA, EPI_LINEAR_ALPHA>(R, A2, Eh, El, A1, p.inv_scale[8], act_scale, neg1,
                                                                   q, out);
        // V: [feature | view embedding] -> 128, relu -> A2[0..3]
        mlp_layer<NP, 9, 8, NERF_F0_V, EPI_RELU>(R, A1, Vh, Vl, A2, p.inv_scale[9], act_scale, neg1, q, out);
        // RGB: 128 -> 3 (4 k-steps), then leave the half-used last chunk
        mlp_layer<NP, 4, 1, NERF_F0_RGB, EPI_RGB>(R, A2, Eh, El, A1, p.inv_scale[10], act_scale, neg1, q, out);
        ring_mid<NP>(R);
        R.pre = read_frag<NP>(ring_next_off<NP>(R.use_off) + lane * 16, 0);
        ring_next<NP>(R);

        if (q == 0) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const unsigned pt = (unsigned)tile * NERF_TILE_PTS + R.wave * NERF_PTS_PER_WAVE + c * 16 + lane;
                if (pt < (unsigned)p.n_pts) {
                    f32x4 r4 = {out.rgb[c][0], out.rgb[c][1], out.rgb[c][2], out.alpha[c]};
                    *reinterpret_cast<f32x4*>(p.raw + (size_t)pt * 4) = r4;
                }
            }
        }
    }
    R2L_WAIT_VMCNT(0);
}

// ------------------------------------------------------------------------------------
// FP16_FP8: the hand-scheduled layer chain.  HIP code computes the tile's embedding fragments and stores raw; all
// eleven layers of the tile are ONE inline-asm block generated by gen/nerf_gen.py (nerf_mlp_asm.inc): fixed register
// map, fp16 main pass + bf6 x bf6 correction terms, three fp16 passes on the embedding k-steps, 4 x 32 KiB LDS ring fed
// by LDS-DMA with counted waits, bias / scale table resident in LDS.  The block owns v0-v247, a0-a95, s40-s55; its
// inputs are the twelve fragment operands (AGPRs), its outputs eight VGPRs: (rgb, sigma) x act_scale of the lane's two
// points, valid in lane quarter 0.  The generator's CPU emulator checks the stream against a float64 network
// (tests/test_nerf_gen_cpu.py).
// ------------------------------------------------------------------------------------
// X1 = true is R2L_PREC_FP16X1: the same chain generated without its correction terms (NERF_GEN_FMT=f16 -> nerf_mlpx_*.inc): one fp16
// pass on the 256-wide sources, the embedding k-steps hi(W) x (hi(E) + lo(E)); 1.30 MB of stream per tile instead of 2.17.
// NC = 3 (X1 only, NERF_GEN_FMT=f16c3 -> nerf_mlpx3_*.inc): three column tiles of 16 points per wave, 192 points per workgroup tile.
// P3 = true (NC = 2) is R2L_PREC_FP16X3_ASM: fp16x3's arithmetic on the chain (NERF_GEN_FMT=f16p3 -> nerf_mlpp3_*.inc): per k-step
// hi(W) hi(a) + hi(W) lo(a) + lo(W) hi(a) on one accumulate chain, lo(a) in a second pair of activation sets (AGPRs), 2.43 MB of
// stream per tile; for teachers whose sharp densities need fp32-grade arithmetic -- every trained one.
// MIX = true (NC = 2) is R2L_PREC_FP16_MIX (round 6): the bf6 chain with trunk layers L1 .. L<NERF_MIX_K> in three fp16 passes
// (NERF_GEN_FMT=mix -> nerf_mlpm_*.inc): for the FINE pass of trained teachers, whose sharp tail amplifies what the early layers get wrong
// (profiles/r06_teacher_mixed_study.txt); lo(a) of those layers' sources in AGPRs the bf6 chain leaves free (a176-a255).
// ALPHA = true (P3 only; round 6): the three-pass chain WITHOUT the view branch (NERF_GEN_FMT=f16p3a -> nerf_mlpp3a_*.inc): raw = (0, 0, 0, sigma).
// The coarse pass of a render whose caller does not take rgb0 (nerf_set_skip_rgb0): sample_pdf and the fine pass see the coarse network
// through its densities only (main.py:716-733), which this build computes bit for bit as the full chain does; 17 % fewer MACs.
// SKIPV = true (P3 or MIX; round 6): the chain with a second exit behind the density (NERF_GEN_FMT=f16p3s / mixs -> nerf_mlpp3s_*.inc /
// nerf_mlpms_*.inc): the alpha row is computed first; when none of the workgroup tile's 128 points has a positive density -- alpha = 0,
// weight 0 exactly, the colour cannot reach rgb_map (main.py:600-606) -- the feature rows, the views layer and the rgb layer are skipped
// and raw = (0, 0, 0, sigma).  The four waves agree through one of two LDS words (alternating per tile) and a barrier.
template <bool X1, int NC, bool P3 = false, bool MIX = false, bool ALPHA = false, bool SKIPV = false>
__global__ __launch_bounds__(256, 1) void nerf_chain_kernel(NerfMlpParams p) {
    static_assert(!ALPHA || P3, "the chain without its view branch exists for the three-pass format");
    static_assert(!SKIPV || ((P3 || MIX) && !ALPHA && NC == 2), "the second exit exists for the three-pass and the mixed chain");
    static_assert(NC == 2 || (X1 && (NC == 3 || NC == 4)), "three / four column tiles exist for the fp16-only chain");
    static_assert(!P3 || (!X1 && NC == 2), "the three-pass chain is a two-column-tile build");
    static_assert(!MIX || (!X1 && !P3 && NC == 2), "the mixed chain is a two-column-tile build of the bf6 chain");
    extern __shared__ __attribute__((aligned(16))) char nerf_chain_lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    {   // resident table: per layer 272 f32 bias (act_scale domain) | E8M0 weight scales (nerf_common.h)
        const uint4* src = reinterpret_cast<const uint4*>(p.wimg + (SKIPV ? (MIX ? NERF_CHAINMS_STREAM_BYTES : NERF_CHAINP3S_STREAM_BYTES) : ALPHA ? NERF_CHAINP3A_STREAM_BYTES : MIX ? NERF_CHAINM_STREAM_BYTES : P3 ? NERF_CHAINP3_STREAM_BYTES : (X1 ? NERF_CHAINX_STREAM_BYTES : NERF_CHAIN_STREAM_BYTES)));
        uint4* dst = reinterpret_cast<uint4*>(nerf_chain_lds + NERF_CHAIN_RING_BYTES);
        for (int i = threadIdx.x; i < NERF_CHAIN_AUX_BYTES / 16; i += 256) dst[i] = src[i];
        if constexpr (SKIPV)
            if (threadIdx.x < 4) reinterpret_cast<unsigned*>(nerf_chain_lds + NERF_CHAIN_LDS)[threadIdx.x] = 0u;     // the second exit's two words
    }
    __syncthreads();
    if constexpr (SKIPV && MIX) {
        asm volatile(
#include "nerf_mlpms_pro_asm.inc"
            :
            : [wimg] "s"(p.wimg), [wave] "s"(wave)
            :
#include "nerf_mlpms_pro_clobbers.inc"
        );
    } else if constexpr (SKIPV) {
        asm volatile(
#include "nerf_mlpp3s_pro_asm.inc"
            :
            : [wimg] "s"(p.wimg), [wave] "s"(wave)
            :
#include "nerf_mlpp3s_pro_clobbers.inc"
        );
    } else if constexpr (ALPHA) {
        asm volatile(
#include "nerf_mlpp3a_pro_asm.inc"
            :
            : [wimg] "s"(p.wimg), [wave] "s"(wave)
            :
#include "nerf_mlpp3a_pro_clobbers.inc"
        );
    } else if constexpr (MIX) {
        asm volatile(
#include "nerf_mlpm_pro_asm.inc"
            :
            : [wimg] "s"(p.wimg), [wave] "s"(wave)
            :
#include "nerf_mlpm_pro_clobbers.inc"
        );
    } else if constexpr (P3) {
        asm volatile(
#include "nerf_mlpp3_pro_asm.inc"
            :
            : [wimg] "s"(p.wimg), [wave] "s"(wave)
            :
#include "nerf_mlpp3_pro_clobbers.inc"
        );
    } else if constexpr (X1 && NC == 4) {
        asm volatile(
#include "nerf_mlpx4_pro_asm.inc"
            :
            : [wimg] "s"(p.wimg), [wave] "s"(wave)
            :
#include "nerf_mlpx4_pro_clobbers.inc"
        );
    } else if constexpr (X1 && NC == 3) {
        asm volatile(
#include "nerf_mlpx3_pro_asm.inc"
            :
            : [wimg] "s"(p.wimg), [wave] "s"(wave)
            :
#include "nerf_mlpx3_pro_clobbers.inc"
        );
    } else if constexpr (X1) {
        asm volatile(
#include "nerf_mlpx_pro_asm.inc"
            :
            : [wimg] "s"(p.wimg), [wave] "s"(wave)
            :
#include "nerf_mlpx_pro_clobbers.inc"
        );
    } else {
        asm volatile(
#include "nerf_mlp_pro_asm.inc"
            :
            : [wimg] "s"(p.wimg), [wave] "s"(wave)
            :
#include "nerf_mlp_pro_clobbers.inc"
        );
    }
    const float inv = 1.0f / p.act_scale;
    // four column tiles leave no register for the given view directions of the NEXT tile (they travel across the asm block): that
    // build takes them as rays_d / |rays_d| only; renders with given directions (NDC) launch the three-tile build (nerf_launch_mlp)
    constexpr bool VD = NC != 4;
    NerfTileRawT<NC, VD> raw;
    if ((int)blockIdx.x < p.n_tiles) nerf_tile_load<NC, VD>(p, blockIdx.x, wave, lane, raw);
    int fl_parity = 0;
    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        f16x8 Eh[2][NC], El[2][NC], Vh[2][NC], Vl[2][NC];
#ifdef NERF_SKIP_EMBED      // diagnostics only (wrong results): what the un-overlapped embedding prologue costs (tools/build_teacher_variant.sh)
        for (int e = 0; e < 2; ++e)
            for (int c = 0; c < NC; ++c)
                for (int j = 0; j < 8; ++j) Eh[e][c][j] = El[e][c][j] = Vh[e][c][j] = Vl[e][c][j] = (f16)(raw.o[c][0] * (float)(e + j));
#else
        nerf_tile_embed<2, NC, VD>(p, raw, lane, Eh, El, Vh, Vl);
#endif
        // the next tile's rays and depths travel while this tile's layers run (the values wait in AGPRs)
        if (tile + (int)gridDim.x < p.n_tiles) nerf_tile_load<NC, VD>(p, tile + gridDim.x, wave, lane, raw);
        float o[4 * NC];
#define NERF_CHAIN_OUT2                                                                                                           \
            : [o0] "=&v"(o[0]), [o1] "=&v"(o[1]), [o2] "=&v"(o[2]), [o3] "=&v"(o[3]), [o4] "=&v"(o[4]), [o5] "=&v"(o[5]),         \
              [o6] "=&v"(o[6]), [o7] "=&v"(o[7])
#define NERF_CHAIN_IN2                                                                                                            \
            : [wimg] "s"(p.wimg), [wave] "s"(wave), [eh00] "a"(Eh[0][0]), [eh01] "a"(Eh[0][1]), [eh10] "a"(Eh[1][0]),             \
              [eh11] "a"(Eh[1][1]), [el00] "a"(El[0][0]), [el01] "a"(El[0][1]), [el10] "a"(El[1][0]), [el11] "a"(El[1][1]),       \
              [vh0] "a"(Vh[0][0]), [vh1] "a"(Vh[0][1]), [vl0] "a"(Vl[0][0]), [vl1] "a"(Vl[0][1])
        // the second exit's LDS word of this tile (its OR lives there; the other word is cleared for the next tile)
        const int fl = NERF_CHAIN_LDS + 4 * fl_parity;
        fl_parity ^= 1;
        if constexpr (SKIPV && MIX) {
            asm volatile(
#include "nerf_mlpms_asm.inc"
                NERF_CHAIN_OUT2
                NERF_CHAIN_IN2, [fl] "s"(fl)
                :
#include "nerf_mlpms_clobbers.inc"
            );
        } else if constexpr (SKIPV) {
            asm volatile(
#include "nerf_mlpp3s_asm.inc"
                NERF_CHAIN_OUT2
                NERF_CHAIN_IN2, [fl] "s"(fl)
                :
#include "nerf_mlpp3s_clobbers.inc"
            );
        } else if constexpr (ALPHA) {
            asm volatile(
#include "nerf_mlpp3a_asm.inc"
                NERF_CHAIN_OUT2
                NERF_CHAIN_IN2
                :
#include "nerf_mlpp3a_clobbers.inc"
            );
        } else if constexpr (MIX) {
            asm volatile(
#include "nerf_mlpm_asm.inc"
                NERF_CHAIN_OUT2
                NERF_CHAIN_IN2
                :
#include "nerf_mlpm_clobbers.inc"
            );
        } else if constexpr (P3) {
            asm volatile(
#include "nerf_mlpp3_asm.inc"
                NERF_CHAIN_OUT2
                NERF_CHAIN_IN2
                :
#include "nerf_mlpp3_clobbers.inc"
            );
        } else if constexpr (X1 && NC == 4) {
            asm volatile(
#include "nerf_mlpx4_asm.inc"
                NERF_CHAIN_OUT2, [o8] "=&v"(o[8]), [o9] "=&v"(o[9]), [o10] "=&v"(o[10]), [o11] "=&v"(o[11]), [o12] "=&v"(o[12]),
                  [o13] "=&v"(o[13]), [o14] "=&v"(o[14]), [o15] "=&v"(o[15])
                NERF_CHAIN_IN2, [eh02] "a"(Eh[0][2]), [eh12] "a"(Eh[1][2]), [el02] "a"(El[0][2]), [el12] "a"(El[1][2]),
                  [vh2] "a"(Vh[0][2]), [vl2] "a"(Vl[0][2]), [eh03] "a"(Eh[0][3]), [eh13] "a"(Eh[1][3]), [el03] "a"(El[0][3]),
                  [el13] "a"(El[1][3]), [vh3] "a"(Vh[0][3]), [vl3] "a"(Vl[0][3])
                :
#include "nerf_mlpx4_clobbers.inc"
            );
        } else if constexpr (X1 && NC == 3) {
            asm volatile(
#include "nerf_mlpx3_asm.inc"
                NERF_CHAIN_OUT2, [o8] "=&v"(o[8]), [o9] "=&v"(o[9]), [o10] "=&v"(o[10]), [o11] "=&v"(o[11])
                NERF_CHAIN_IN2, [eh02] "a"(Eh[0][2]), [eh12] "a"(Eh[1][2]), [el02] "a"(El[0][2]), [el12] "a"(El[1][2]),
                  [vh2] "a"(Vh[0][2]), [vl2] "a"(Vl[0][2])
                :
#include "nerf_mlpx3_clobbers.inc"
            );
        } else if constexpr (X1) {
            asm volatile(
#include "nerf_mlpx_asm.inc"
                NERF_CHAIN_OUT2
                NERF_CHAIN_IN2
                :
#include "nerf_mlpx_clobbers.inc"
            );
        } else {
            asm volatile(
#include "nerf_mlp_asm.inc"
                NERF_CHAIN_OUT2
                NERF_CHAIN_IN2
                :
#include "nerf_mlp_clobbers.inc"
            );
        }
#undef NERF_CHAIN_OUT2
#undef NERF_CHAIN_IN2
        if (lane < 16) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const unsigned pt = (unsigned)tile * (64 * NC) + wave * (16 * NC) + c * 16 + lane;
                if (pt < (unsigned)p.n_pts)
                    *reinterpret_cast<f32x4*>(p.raw + (size_t)pt * 4) = f32x4{o[4 * c] * inv, o[4 * c + 1] * inv, o[4 * c + 2] * inv, o[4 * c + 3] * inv};
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last block's refill of the ring
}

// R2L_PREC_FP16X1 without given view directions (round 5): the four-column-tile chain as ONE generated statement that holds the tile
// loop (gen/nerf_gen.py NERF_GEN_FMT=f16c4e -> nerf_mlpx4e_asm.inc).  What nerf_chain_kernel<true, 4> does in HIP code between two
// blocks while the matrix pipe idles -- nerf_tile_load, nerf_tile_embed, the raw stores: 2.3 ms of a 33.4 ms frame -- is part of the
// stream here: 12 global loads into AGPRs at the start of a block, the NEXT tile's embedding as filler instructions in the shadow
// of the MFMAs of L6 .. V (operation for operation nerf_tile_embed's arithmetic: the results are bitwise those of the other builds,
// tests/test_teacher_watch_gpu.py), four masked global_store_dwordx4 at its end.  The statement owns every VGPR / AGPR and s40-s95.
__global__ __launch_bounds__(256, 1) void nerf_chain_emb_kernel(NerfMlpParams p) {
    extern __shared__ __attribute__((aligned(16))) char nerf_chain_lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    {   // resident table: per layer 272 f32 bias (act_scale domain) (nerf_common.h)
        const uint4* src = reinterpret_cast<const uint4*>(p.wimg + NERF_CHAINX_STREAM_BYTES);
        uint4* dst = reinterpret_cast<uint4*>(nerf_chain_lds + NERF_CHAIN_RING_BYTES);
        for (int i = threadIdx.x; i < NERF_CHAIN_AUX_BYTES / 16; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    asm volatile(
#include "nerf_mlpx4e_asm.inc"
        :
        : [wimg] "s"(p.wimg), [wave] "s"(wave), [ro] "s"(p.rays_o), [rd] "s"(p.rays_d), [z] "s"(p.z), [raw] "s"(p.raw),
          [npts] "s"((unsigned)p.n_pts), [S] "s"((unsigned)p.S), [zs] "s"((unsigned)p.z_stride), [magic] "s"(p.div_magic),
          [sh1] "s"(p.div_sh1), [sh2] "s"(p.div_sh2), [tile] "s"((unsigned)blockIdx.x), [grid] "s"((unsigned)gridDim.x),
          [ntiles] "s"((unsigned)p.n_tiles)
        :
#include "nerf_mlpx4e_clobbers.inc"
    );
}

// ====================================================================================
// get_rays
// ====================================================================================
__global__ void nerf_get_rays_kernel(float c00, float c01, float c02, float c03, float c10, float c11, float c12,
                                     float c13, float c20, float c21, float c22, float c23, int W, float half_w,
                                     float half_h, float focal, int pix_begin, int n, float* __restrict__ rays_o,
                                     float* __restrict__ rays_d) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int pix = pix_begin + i;
    int jrow = pix / W, icol = pix - jrow * W;
    float dx = __fdiv_rn((float)icol - half_w, focal);
    float dy = -__fdiv_rn((float)jrow - half_h, focal);
    const float c[12] = {c00, c01, c02, c03, c10, c11, c12, c13, c20, c21, c22, c23};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float s = __fadd_rn(__fmul_rn(dx, c[4 * k + 0]), __fmul_rn(dy, c[4 * k + 1]));
        rays_d[(size_t)i * 3 + k] = __fadd_rn(s, __fmul_rn(-1.0f, c[4 * k + 2]));
        rays_o[(size_t)i * 3 + k] = c[4 * k + 3];
    }
}

// ====================================================================================
// ndc_rays (utils/run_nerf_raybased_helpers.py:260-279) + the view directions render() takes from
// the world-space rays before the projection (main.py:148-157): one thread per ray, the
// reference's operation order, one rounding per operation.
//   c0 = fl32(-1/(W/(2 focal))), c1 = fl32(-1/(H/(2 focal))), two_near = fl32(2 near), mtwo_near = fl32(-2 near)
// ====================================================================================
__global__ void nerf_ndc_rays_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d, int n, float c0,
                                     float c1, float near_, float two_near, float mtwo_near, float* __restrict__ out_o,
                                     float* __restrict__ out_d, float* __restrict__ viewdirs) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float ox = rays_o[(size_t)i * 3], oy = rays_o[(size_t)i * 3 + 1], oz = rays_o[(size_t)i * 3 + 2];
    const float dx = rays_d[(size_t)i * 3], dy = rays_d[(size_t)i * 3 + 1], dz = rays_d[(size_t)i * 3 + 2];
    if (viewdirs) {  // rays_d / ||rays_d|| of the WORLD rays (same arithmetic as nerf_mlp_kernel)
        const float nrm = sqrtf(__fadd_rn(__fadd_rn(dx * dx, dy * dy), dz * dz));
        viewdirs[(size_t)i * 3] = __fdiv_rn(dx, nrm);
        viewdirs[(size_t)i * 3 + 1] = __fdiv_rn(dy, nrm);
        viewdirs[(size_t)i * 3 + 2] = __fdiv_rn(dz, nrm);
    }
    if (!out_o) return;
    const float t = __fdiv_rn(-__fadd_rn(near_, oz), dz);       // -(near + o_z) / d_z
    const float px = __fadd_rn(ox, __fmul_rn(t, dx));            // origin moved to the near plane
    const float py = __fadd_rn(oy, __fmul_rn(t, dy));
    const float pz = __fadd_rn(oz, __fmul_rn(t, dz));
    const float rx = __fdiv_rn(px, pz), ry = __fdiv_rn(py, pz);  // used by the d0 / d1 terms
    out_o[(size_t)i * 3] = __fdiv_rn(__fmul_rn(c0, px), pz);     // (c0 * o_x) / o_z
    out_o[(size_t)i * 3 + 1] = __fdiv_rn(__fmul_rn(c1, py), pz);
    out_o[(size_t)i * 3 + 2] = __fadd_rn(1.0f, __fdiv_rn(two_near, pz));
    out_d[(size_t)i * 3] = __fmul_rn(c0, __fadd_rn(__fdiv_rn(dx, dz), -rx));
    out_d[(size_t)i * 3 + 1] = __fmul_rn(c1, __fadd_rn(__fdiv_rn(dy, dz), -ry));
    out_d[(size_t)i * 3 + 2] = __fdiv_rn(mtwo_near, pz);
}

// ====================================================================================
// wave-level scans (64 lanes)
// ====================================================================================
__device__ __forceinline__ double shfl_up_f64(double v, int delta) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_up(lo, delta, 64);
    hi = __shfl_up(hi, delta, 64);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, mask, 64);
    hi = __shfl_xor(hi, mask, 64);
    return __hiloint2double(hi, lo);
}
// inclusive product / sum scans across the wave in double precision (torch's CPU cumprod /
// cumsum accumulate float tensors in double and round each output to float)
__device__ __forceinline__ double wave_scan_mul(double v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        double u = shfl_up_f64(v, d);
        if (lane >= d) v *= u;
    }
    return v;
}
__device__ __forceinline__ double wave_scan_add(double v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        double u = shfl_up_f64(v, d);
        if (lane >= d) v += u;
    }
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_f64(v, m);
    return v;
}

// ====================================================================================
// raw2outputs: one wave per ray, C consecutive samples per lane
// ====================================================================================
template <int C>
__global__ __launch_bounds__(256) void nerf_raw2outputs_kernel(const float* __restrict__ raw,
                                                               const float* __restrict__ z, int z_stride,
                                                               const float* __restrict__ rays_d, int n, int S,
                                                               int white_bkgd, float* __restrict__ rgb_map,
                                                               float* __restrict__ disp_map,
                                                               float* __restrict__ acc_map,
                                                               float* __restrict__ weights_out,
                                                               float* __restrict__ depth_map,
                                                               const float* __restrict__ noise) {
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (ray >= n) return;
    const float* zr = z + (size_t)ray * z_stride;
    const float dx = rays_d[(size_t)ray * 3 + 0], dy = rays_d[(size_t)ray * 3 + 1], dz = rays_d[(size_t)ray * 3 + 2];
    const float norm = sqrtf(__fadd_rn(__fadd_rn(dx * dx, dy * dy), dz * dz));  // torch.norm(rays_d, dim=-1)
    float alpha[C], zi[C], cr[C], cg[C], cb[C];
    double lp[C];
    double prod = 1.0;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = lane * C + c;
        const bool ok = i < S;
        const int ii = ok ? i : S - 1;
        const float zc = zr[ii];
        const float zn = zr[ii + 1 < S ? ii + 1 : S - 1];
        float dist = (ii < S - 1) ? (zn - zc) : 1e10f;  // dists = cat(z[1:]-z[:-1], 1e10)
        dist = dist * norm;
        const f32x4 r4 = *reinterpret_cast<const f32x4*>(raw + ((size_t)ray * S + ii) * 4);
        // raw2alpha(raw[..., 3] + noise, dists), noise = randn * raw_noise_std drawn by the caller (main.py:592-600)
        const float sig = fmaxf(noise ? r4[3] + noise[(size_t)ray * S + ii] : r4[3], 0.0f);  // F.relu
        float a = 1.0f - expf(-sig * dist);                   // 1 - exp(-relu(raw) * dists)
        if (!ok) a = 0.0f;
        alpha[c] = a;
        zi[c] = zc;
        cr[c] = 1.0f / (1.0f + expf(-r4[0]));                 // torch.sigmoid
        cg[c] = 1.0f / (1.0f + expf(-r4[1]));
        cb[c] = 1.0f / (1.0f + expf(-r4[2]));
        lp[c] = prod;                                          // exclusive product inside the lane
        const float pterm = ok ? ((1.0f - a) + 1e-10f) : 1.0f; // 1 - alpha + 1e-10, float ops
        prod *= (double)pterm;
    }
    const double incl = wave_scan_mul(prod, lane);
    double excl = shfl_up_f64(incl, 1);
    if (lane == 0) excl = 1.0;
    float sr = 0.f, sg = 0.f, sb = 0.f, sd = 0.f, sa = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = lane * C + c;
        const float T = (float)(excl * lp[c]);  // cumprod accumulated in double, rounded per element
        const float w = alpha[c] * T;
        if (i < S) {
            if (weights_out) weights_out[(size_t)ray * S + i] = w;
            sr += w * cr[c];
            sg += w * cg[c];
            sb += w * cb[c];
            sd += w * zi[c];
            sa += w;
        }
    }
    sr = wave_sum(sr);
    sg = wave_sum(sg);
    sb = wave_sum(sb);
    sd = wave_sum(sd);
    sa = wave_sum(sa);
    if (lane == 0) {
        if (white_bkgd) {
            const float bg = 1.0f - sa;
            sr += bg;
            sg += bg;
            sb += bg;
        }
        if (rgb_map) {
            rgb_map[(size_t)ray * 3 + 0] = sr;
            rgb_map[(size_t)ray * 3 + 1] = sg;
            rgb_map[(size_t)ray * 3 + 2] = sb;
        }
        if (depth_map) depth_map[ray] = sd;
        if (acc_map) acc_map[ray] = sa;
        if (disp_map) {
            const float q = sd / sa;
            // torch.max(1e-10, q) propagates NaN (0/0 for empty rays), fmaxf would not
            const float m = (q != q) ? q : fmaxf(1e-10f, q);
            disp_map[ray] = 1.0f / m;
        }
    }
}

// ====================================================================================
// sample_pdf: one wave per ray, n_bins <= 64  (utils/run_nerf_raybased_helpers.py:283-330; the reference runs it
// on the CPU, main.py:723-728, so the float accumulation orders below are ATen's CPU orders)
// ====================================================================================
// torch.sum(x, -1) of a contiguous float row on the CPU (ATen SumKernel, vectorized_inner_sum): 8-lane vectors,
// 4 interleaved vector accumulators over groups of 4 vectors, the remaining whole vectors into accumulator 0,
// ((a0 + a1) + a2) + a3 per lane, then a scalar chain: the row's tail elements first, then the 8 lane sums.
// Verified bit for bit against torch.sum on rows of 30 .. 190 floats (tools/torch_sum_order.py).
__device__ __forceinline__ float torch_cpu_row_sum(const float* row /* LDS, zero padded to 64 */, float* cols /* LDS [8] */,
                                                   int nw, int lane) {
    const int nv = nw >> 3, full = (nv >> 2) << 2;
    if (lane < 8) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int v = 0; v < full; v += 4) {
            a0 = __fadd_rn(a0, row[(v + 0) * 8 + lane]);
            a1 = __fadd_rn(a1, row[(v + 1) * 8 + lane]);
            a2 = __fadd_rn(a2, row[(v + 2) * 8 + lane]);
            a3 = __fadd_rn(a3, row[(v + 3) * 8 + lane]);
        }
        for (int v = full; v < nv; ++v) a0 = __fadd_rn(a0, row[v * 8 + lane]);
        cols[lane] = __fadd_rn(__fadd_rn(__fadd_rn(a0, a1), a2), a3);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    float fin = 0.f;
    for (int k = nv * 8; k < nw; ++k) fin = __fadd_rn(fin, row[k]);
    for (int k = 0; k < 8; ++k) fin = __fadd_rn(fin, cols[k]);
    return fin;
}

// bins: [n, n_bins] (stride 0 = one shared row); with bins_are_z the row holds n_bins + 1 depths and
// bins = .5 * (z[1:] + z[:-1]) (main.py:722).  u: nullptr = torch.linspace(0, 1, N) (det, evaluated here by the
// scalar formula), else [N] (u_stride 0) or one row per ray (u_stride = N: the perturb > 0 path draws u per ray).
// cdf_out [n, n_bins], inds_out [n, N] (searchsorted(cdf, u, right=True)): optional parity taps.
__global__ __launch_bounds__(256) void nerf_sample_pdf_kernel(const float* __restrict__ bins, int bins_stride, int bins_are_z,
                                                              const float* __restrict__ weights, int w_stride,
                                                              int w_off, int n, int n_bins,
                                                              const float* __restrict__ u_arr, int u_stride, int N,
                                                              float* __restrict__ samples, float* __restrict__ cdf_out,
                                                              int* __restrict__ inds_out) {
    __shared__ float s_cdf[4][64];
    __shared__ float s_bins[4][64];
    __shared__ float s_w[4][64];
    __shared__ float s_col[4][8];
    const int wv = threadIdx.x >> 6;
    const int ray = blockIdx.x * 4 + wv;
    const int lane = threadIdx.x & 63;
    if (ray >= n) return;
    const int nw = n_bins - 1;
    float w = 0.0f;
    if (lane < nw) w = weights[(size_t)ray * w_stride + w_off + lane] + 1e-5f;  // weights + 1e-5
    s_w[wv][lane] = w;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const float total = torch_cpu_row_sum(s_w[wv], s_col[wv], nw, lane);          // torch.sum(weights, -1)
    const float pdf = (lane < nw) ? w / total : 0.0f;
    const double cs = wave_scan_add((double)pdf, lane);                          // cumsum (double accumulate)
    if (lane < nw) s_cdf[wv][lane + 1] = (float)cs;
    if (lane == 63) s_cdf[wv][0] = 0.0f;                                          // cat(zeros, cdf)
    if (lane < n_bins) {
        const float* br = bins + (size_t)ray * bins_stride;
        s_bins[wv][lane] = bins_are_z ? 0.5f * __fadd_rn(br[lane + 1], br[lane]) : br[lane];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (cdf_out && lane < n_bins) cdf_out[(size_t)ray * n_bins + lane] = s_cdf[wv][lane];
    const float step = N > 1 ? 1.0f / (float)(N - 1) : 0.0f;
    for (int k = lane; k < N; k += 64) {
        float u;
        if (u_arr) u = u_arr[(size_t)ray * u_stride + k];
        else u = (k < N / 2 || N == 1) ? __fmul_rn(step, (float)k) : __fsub_rn(1.0f, __fmul_rn(step, (float)(N - k - 1)));
        int lo = 0, hi = n_bins;  // searchsorted(cdf, u, right=True): first index with cdf > u
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_cdf[wv][mid] <= u) lo = mid + 1; else hi = mid;
        }
        if (inds_out) inds_out[(size_t)ray * N + k] = lo;
        const int below = lo - 1 > 0 ? lo - 1 : 0;
        const int above = lo < n_bins - 1 ? lo : n_bins - 1;
        const float c0 = s_cdf[wv][below], c1 = s_cdf[wv][above];
        const float b0 = s_bins[wv][below], b1 = s_bins[wv][above];
        float denom = c1 - c0;
        if (denom < 1e-5f) denom = 1.0f;
        const float t = (u - c0) / denom;
        samples[(size_t)ray * N + k] = b0 + t * (b1 - b0);
    }
}

// rows of N <= 256 floats sorted ascending (bitonic network in LDS, one wave per row): with random uniforms
// (perturb > 0) sample_pdf's output is not monotone and the reference's sort (main.py:730-732) is a real sort
__global__ __launch_bounds__(256) void nerf_sort_rows_kernel(const float* __restrict__ x, int n, int N, float* __restrict__ out) {
    __shared__ float s[4][256];
    const int wv = threadIdx.x >> 6;
    const int ray = blockIdx.x * 4 + wv;
    const int lane = threadIdx.x & 63;
    if (ray >= n) return;
    for (int i = lane; i < 256; i += 64) s[wv][i] = i < N ? x[(size_t)ray * N + i] : __builtin_huge_valf();
    for (int k = 2; k <= 256; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (int i = lane; i < 256; i += 64) {
                const int p = i ^ j;
                if (p > i) {
                    const float a = s[wv][i], b = s[wv][p];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) {
                        s[wv][i] = b;
                        s[wv][p] = a;
                    }
                }
            }
        }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int i = lane; i < N; i += 64) out[(size_t)ray * N + i] = s[wv][i];
}

// torch.std(z_samples, dim=-1, unbiased=False) of [n, N] rows (main.py:749): one wave per ray
__global__ __launch_bounds__(256) void nerf_row_std_kernel(const float* __restrict__ x, int n, int N, float* __restrict__ out) {
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (ray >= n) return;
    double s = 0.0;
    for (int k = lane; k < N; k += 64) s += (double)x[(size_t)ray * N + k];
    const double mean = wave_sum_f64(s) / N;
    double q = 0.0;
    for (int k = lane; k < N; k += 64) {
        const double d = (double)x[(size_t)ray * N + k] - mean;
        q += d * d;
    }
    const double var = wave_sum_f64(q) / N;
    if (lane == 0) out[ray] = (float)sqrt(var);
}

// ====================================================================================
// merge of two ascending rows (the reference sorts the concatenation): one wave per ray
// ====================================================================================
__global__ __launch_bounds__(256) void nerf_merge_kernel(const float* __restrict__ a, int a_stride, int na,
                                                         const float* __restrict__ b, int nb, int n,
                                                         float* __restrict__ out) {
    __shared__ float s_a[4][256];
    __shared__ float s_b[4][256];
    const int wv = threadIdx.x >> 6;
    const int ray = blockIdx.x * 4 + wv;
    const int lane = threadIdx.x & 63;
    if (ray >= n) return;
    for (int i = lane; i < na; i += 64) s_a[wv][i] = a[(size_t)ray * a_stride + i];
    for (int i = lane; i < nb; i += 64) s_b[wv][i] = b[(size_t)ray * nb + i];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    float* o = out + (size_t)ray * (na + nb);
    for (int i = lane; i < na; i += 64) {  // rank of a[i] = i + #{b < a[i]}
        const float v = s_a[wv][i];
        int lo = 0, hi = nb;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_b[wv][mid] < v) lo = mid + 1; else hi = mid;
        }
        o[i + lo] = v;
    }
    for (int j = lane; j < nb; j += 64) {  // rank of b[j] = j + #{a <= b[j]}
        const float v = s_b[wv][j];
        int lo = 0, hi = na;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_a[wv][mid] <= v) lo = mid + 1; else hi = mid;
        }
        o[j + lo] = v;
    }
}

// ====================================================================================
// coarse pass -> fine depths in ONE launch (main.py:705-732 on the deterministic test path): raw2outputs of the coarse raw,
// sample_pdf on weights[..., 1:-1] over the midpoints of z, and the merge of z with the samples -- one wave per ray, the weights,
// the cdf and the samples never leave LDS (stand-alone: three launches, weights and samples through HBM twice).  The arithmetic
// is the three kernels' above, statement for statement (same float / double orders: the results are bit-identical, asserted by
// tests/test_teacher_gpu.py); S0 <= 64 coarse samples, N <= 256 fine ones, u one shared row (per-ray uniforms need the sort).
// ====================================================================================
__global__ __launch_bounds__(256) void nerf_coarse_scan_kernel(const float* __restrict__ raw, const float* __restrict__ z, int z_stride,
                                                               const float* __restrict__ rays_d, int n, int S, int white_bkgd,
                                                               const float* __restrict__ noise, const float* __restrict__ u_arr, int N,
                                                               float* __restrict__ rgb_map, float* __restrict__ disp_map,
                                                               float* __restrict__ acc_map, float* __restrict__ samples,
                                                               float* __restrict__ z_all) {
    __shared__ float s_z[4][64];
    __shared__ float s_wt[4][64];
    __shared__ float s_w[4][64];
    __shared__ float s_cdf[4][64];
    __shared__ float s_bins[4][64];
    __shared__ float s_col[4][8];
    __shared__ float s_s[4][256];
    const int wv = threadIdx.x >> 6;
    const int ray = blockIdx.x * 4 + wv;
    const int lane = threadIdx.x & 63;
    if (ray >= n) return;
    // ---- raw2outputs, C = 1 (nerf_raw2outputs_kernel<1>)
    const float* zr = z + (size_t)ray * z_stride;
    const float dx = rays_d[(size_t)ray * 3 + 0], dy = rays_d[(size_t)ray * 3 + 1], dz = rays_d[(size_t)ray * 3 + 2];
    const float norm = sqrtf(__fadd_rn(__fadd_rn(dx * dx, dy * dy), dz * dz));
    const bool ok = lane < S;
    const int ii = ok ? lane : S - 1;
    const float zc = zr[ii];
    const float zn = zr[ii + 1 < S ? ii + 1 : S - 1];
    float dist = (ii < S - 1) ? (zn - zc) : 1e10f;
    dist = dist * norm;
    const f32x4 r4 = *reinterpret_cast<const f32x4*>(raw + ((size_t)ray * S + ii) * 4);
    const float sig = fmaxf(noise ? r4[3] + noise[(size_t)ray * S + ii] : r4[3], 0.0f);
    float a = 1.0f - expf(-sig * dist);
    if (!ok) a = 0.0f;
    const float cr = 1.0f / (1.0f + expf(-r4[0])), cg = 1.0f / (1.0f + expf(-r4[1])), cb = 1.0f / (1.0f + expf(-r4[2]));
    const float pterm = ok ? ((1.0f - a) + 1e-10f) : 1.0f;
    const double incl = wave_scan_mul((double)pterm, lane);
    double excl = shfl_up_f64(incl, 1);
    if (lane == 0) excl = 1.0;
    const float T = (float)(excl * 1.0);
    const float wgt = a * T;
    float sr = 0.f, sg = 0.f, sb = 0.f, sd = 0.f, sa = 0.f;
    if (ok) {
        sr = wgt * cr;
        sg = wgt * cg;
        sb = wgt * cb;
        sd = wgt * zc;
        sa = wgt;
    }
    s_wt[wv][lane] = ok ? wgt : 0.0f;
    s_z[wv][lane] = zc;
    sr = wave_sum(sr);
    sg = wave_sum(sg);
    sb = wave_sum(sb);
    sd = wave_sum(sd);
    sa = wave_sum(sa);
    if (lane == 0) {
        if (white_bkgd) {
            const float bg = 1.0f - sa;
            sr += bg;
            sg += bg;
            sb += bg;
        }
        if (rgb_map) {
            rgb_map[(size_t)ray * 3 + 0] = sr;
            rgb_map[(size_t)ray * 3 + 1] = sg;
            rgb_map[(size_t)ray * 3 + 2] = sb;
        }
        if (acc_map) acc_map[ray] = sa;
        if (disp_map) {
            const float q = sd / sa;
            const float m = (q != q) ? q : fmaxf(1e-10f, q);
            disp_map[ray] = 1.0f / m;
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // ---- sample_pdf(z_mid, weights[..., 1:-1], N, det) (nerf_sample_pdf_kernel with bins_are_z, w_off = 1)
    const int n_bins = S - 1, nw = n_bins - 1;
    float w = 0.0f;
    if (lane < nw) w = s_wt[wv][1 + lane] + 1e-5f;
    s_w[wv][lane] = w;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const float total = torch_cpu_row_sum(s_w[wv], s_col[wv], nw, lane);
    const float pdf = (lane < nw) ? w / total : 0.0f;
    const double cs = wave_scan_add((double)pdf, lane);
    if (lane < nw) s_cdf[wv][lane + 1] = (float)cs;
    if (lane == 63) s_cdf[wv][0] = 0.0f;
    if (lane < n_bins) s_bins[wv][lane] = 0.5f * __fadd_rn(s_z[wv][lane + 1], s_z[wv][lane]);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const float step = N > 1 ? 1.0f / (float)(N - 1) : 0.0f;
    for (int k = lane; k < N; k += 64) {
        float u;
        if (u_arr) u = u_arr[k];
        else u = (k < N / 2 || N == 1) ? __fmul_rn(step, (float)k) : __fsub_rn(1.0f, __fmul_rn(step, (float)(N - k - 1)));
        int lo = 0, hi = n_bins;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_cdf[wv][mid] <= u) lo = mid + 1; else hi = mid;
        }
        const int below = lo - 1 > 0 ? lo - 1 : 0;
        const int above = lo < n_bins - 1 ? lo : n_bins - 1;
        const float c0 = s_cdf[wv][below], c1 = s_cdf[wv][above];
        const float b0 = s_bins[wv][below], b1 = s_bins[wv][above];
        float denom = c1 - c0;
        if (denom < 1e-5f) denom = 1.0f;
        const float t = (u - c0) / denom;
        const float smp = b0 + t * (b1 - b0);
        s_s[wv][k] = smp;
        samples[(size_t)ray * N + k] = smp;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // ---- sort(cat(z, samples)) of two ascending rows (nerf_merge_kernel)
    float* o = z_all + (size_t)ray * (S + N);
    if (lane < S) {
        const float v = s_z[wv][lane];
        int lo = 0, hi = N;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_s[wv][mid] < v) lo = mid + 1; else hi = mid;
        }
        o[lane + lo] = v;
    }
    for (int j = lane; j < N; j += 64) {
        const float v = s_s[wv][j];
        int lo = 0, hi = S;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_z[wv][mid] <= v) lo = mid + 1; else hi = mid;
        }
        o[j + lo] = v;
    }
}

// ====================================================================================
// launchers
// ====================================================================================
template <typename K>
static hipError_t launch_big_lds(K kernel, std::atomic<bool>* attr_set, int lds, const NerfMlpParams& p, int grid,
                                 hipStream_t stream) {
    // the > 64 KiB dynamic-LDS opt-in is per device: a process may drive several GPUs
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, stream, p);
    return hipGetLastError();
}

hipError_t nerf_launch_mlp(const NerfMlpParams& p, int mode, int grid, hipStream_t stream, int x1_col_tiles, bool stream_embed, bool alpha_only,
                           bool second_exit) {
    static std::atomic<bool> attr_set[11][64];  // zero-initialised; the opt-in call itself is idempotent
    if (second_exit && mode == R2L_PREC_FP16X3_ASM)
        return launch_big_lds(&nerf_chain_kernel<false, 2, true, false, false, true>, attr_set[9], NERF_CHAIN_LDS_SKIP, p, grid, stream);
    if (second_exit && mode == R2L_PREC_FP16_MIX)
        return launch_big_lds(&nerf_chain_kernel<false, 2, false, true, false, true>, attr_set[10], NERF_CHAIN_LDS_SKIP, p, grid, stream);
    if (alpha_only && mode == R2L_PREC_FP16X3_ASM)
        return launch_big_lds(&nerf_chain_kernel<false, 2, true, false, true>, attr_set[8], NERF_CHAIN_LDS, p, grid, stream);
    if (mode == R2L_PREC_FP16_MIX) return launch_big_lds(&nerf_chain_kernel<false, 2, false, true>, attr_set[7], NERF_CHAIN_LDS, p, grid, stream);
    if (mode == R2L_PREC_FP16X3_ASM) return launch_big_lds(&nerf_chain_kernel<false, 2, true>, attr_set[6], NERF_CHAIN_LDS, p, grid, stream);
    if (mode == R2L_PREC_FP16X1 && x1_col_tiles == 4 && stream_embed)
        return launch_big_lds(&nerf_chain_emb_kernel, attr_set[5], NERF_CHAIN_LDS, p, grid, stream);
    if (mode == R2L_PREC_FP16_FP8) return launch_big_lds(&nerf_chain_kernel<false, 2>, attr_set[0], NERF_CHAIN_LDS, p, grid, stream);
    if (mode == R2L_PREC_FP16X3) return launch_big_lds(&nerf_mlp_kernel<2>, attr_set[1], KCfg<2>::LDS, p, grid, stream);
    // FP16X1: the generated chain without correction terms (round 4; the compiler-scheduled nerf_mlp_kernel<1> it replaces: 49.5 ms per
    // frame), with three column tiles per wave (192-point workgroup tiles: p.n_tiles counts those) or two
    if (x1_col_tiles == 4) return launch_big_lds(&nerf_chain_kernel<true, 4>, attr_set[4], NERF_CHAIN_LDS, p, grid, stream);
    if (x1_col_tiles == 3) return launch_big_lds(&nerf_chain_kernel<true, 3>, attr_set[3], NERF_CHAIN_LDS, p, grid, stream);
    return launch_big_lds(&nerf_chain_kernel<true, 2>, attr_set[2], NERF_CHAIN_LDS, p, grid, stream);
}

hipError_t nerf_launch_ndc_rays(const float* rays_o, const float* rays_d, int n, int H, int W, double focal, float near_,
                                float* out_o, float* out_d, float* viewdirs, hipStream_t stream) {
    const float c0 = (float)(-1. / (W / (2. * focal))), c1 = (float)(-1. / (H / (2. * focal)));
    hipLaunchKernelGGL(nerf_ndc_rays_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, rays_o, rays_d, n, c0, c1, near_,
                       (float)(2. * near_), (float)(-2. * near_), out_o, out_d, viewdirs);
    return hipGetLastError();
}

hipError_t nerf_launch_get_rays(const float* c, int W, float half_w, float half_h, float focal, int pix_begin,
                                int n, float* rays_o, float* rays_d, hipStream_t stream) {
    hipLaunchKernelGGL(nerf_get_rays_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, c[0], c[1], c[2], c[3],
                       c[4], c[5], c[6], c[7], c[8], c[9], c[10], c[11], W, half_w, half_h, focal, pix_begin, n,
                       rays_o, rays_d);
    return hipGetLastError();
}

hipError_t nerf_launch_raw2outputs(const float* raw, const float* z, int z_stride, const float* rays_d, int n,
                                   int S, int white_bkgd, float* rgb, float* disp, float* acc, float* weights,
                                   float* depth, hipStream_t stream, const float* noise) {
    const dim3 grid((n + 3) / 4), block(256);
    const int C = (S + 63) / 64;
#define R2O(c)                                                                                              \
    hipLaunchKernelGGL(nerf_raw2outputs_kernel<c>, grid, block, 0, stream, raw, z, z_stride, rays_d, n, S, \
                       white_bkgd, rgb, disp, acc, weights, depth, noise)
    switch (C) {
        case 1: R2O(1); break;
        case 2: R2O(2); break;
        case 3: R2O(3); break;
        case 4: R2O(4); break;
        default: return hipErrorInvalidValue;
    }
#undef R2O
    return hipGetLastError();
}

hipError_t nerf_launch_sample_pdf(const float* bins, int bins_stride, int bins_are_z, const float* weights, int w_stride,
                                  int w_off, int n, int n_bins, const float* u, int u_stride, int N, float* samples,
                                  float* cdf_out, int* inds_out, hipStream_t stream) {
    if (n_bins < 2 || n_bins > 64 || N < 1) return hipErrorInvalidValue;  // the reference's sample_pdf itself fails on a single bin
    hipLaunchKernelGGL(nerf_sample_pdf_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, bins, bins_stride, bins_are_z,
                       weights, w_stride, w_off, n, n_bins, u, u_stride, N, samples, cdf_out, inds_out);
    return hipGetLastError();
}

hipError_t nerf_launch_coarse_scan(const float* raw, const float* z, int z_stride, const float* rays_d, int n, int S, int white_bkgd,
                                   const float* noise, const float* u, int N, float* rgb, float* disp, float* acc, float* samples,
                                   float* z_all, hipStream_t stream) {
    if (S < 3 || S > 64 || N < 1 || N > 256) return hipErrorInvalidValue;
    hipLaunchKernelGGL(nerf_coarse_scan_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, raw, z, z_stride, rays_d, n, S, white_bkgd,
                       noise, u, N, rgb, disp, acc, samples, z_all);
    return hipGetLastError();
}

hipError_t nerf_launch_sort_rows(const float* x, int n, int N, float* out, hipStream_t stream) {
    if (N < 1 || N > 256) return hipErrorInvalidValue;
    hipLaunchKernelGGL(nerf_sort_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, x, n, N, out);
    return hipGetLastError();
}

hipError_t nerf_launch_row_std(const float* x, int n, int N, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(nerf_row_std_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, x, n, N, out);
    return hipGetLastError();
}

hipError_t nerf_launch_merge(const float* a, int a_stride, int na, const float* b, int nb, int n, float* out,
                             hipStream_t stream) {
    if (na > 256 || nb > 256 || na < 0 || nb < 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(nerf_merge_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, a, a_stride, na, b, nb, n, out);
    return hipGetLastError();
}
