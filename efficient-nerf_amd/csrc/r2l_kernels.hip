// R2L (neural light field) hot path for MI355X / gfx950, hand-written HIP.
//
//   r2l_resmlp_kernel<NP>
//                          K1+K2+K3 fused: get_rays + 16-point sampling + sinusoidal
//                          embedding + 88-layer width-256 residual MLP + sigmoid, one
//                          persistent workgroup per CU, activations resident in registers,
//                          weights streamed global -> LDS ring (LDS-DMA) -> MFMA A operand.
//                          <2>: fp16 hi/lo split, 3 MFMA per k-step (fp32-grade result, R2L_PREC_FP16X3)
//                          <1>: single fp16 pass (R2L_PREC_FP16X1)
//                          <2, true>: the head layer only, its output left in HBM for the hand-scheduled
//                          body kernel of R2L_PREC_FP16_FP8 (r2l_body.hip) and the tail kernel.
//   r2l_sample_embed_kernel / r2l_embed_kernel
//                          stand-alone K1+K2 (PointSampler.sample_test,
//                          PositionalEmbedder.__call__) for parity tests and the API mirror.
//
// Reference semantics restated here (file:line in MingSun-Tse/Efficient-NeRF):
//   dirs / rays      model/nerf_raybased.py:80-99  == utils/run_nerf_raybased_helpers.py:233-247
//   points           model/nerf_raybased.py:100-102
//   embedding        model/nerf_raybased.py:198-208
//   network          model/nerf_raybased.py:443-465 (ResMLP), :539-544 (NeRF_v3_2.forward)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/r2l_hip.h"
#include "r2l_common.h"
#include "r2l_kernels.h"

#include "r2l_device.h"

// camera ray of flat ray index `ray` (model/nerf_raybased.py:84-99), returned by value with
// scalar members so it lives in registers (arrays handed around by pointer ended up in scratch,
// and every scratch reload drains the LDS-DMA ring with a vmcnt(0)).
struct Ray6 {
    float ox, oy, oz, dx, dy, dz;
    __device__ __forceinline__ float o(int k) const { return k == 0 ? ox : (k == 1 ? oy : oz); }
    __device__ __forceinline__ float d(int k) const { return k == 0 ? dx : (k == 1 ? dy : dz); }
};

__device__ __forceinline__ Ray6 make_ray(const R2LParams& p, int ray) {
    Ray6 r;
    if (p.rays_o != nullptr) {
        const float* po = p.rays_o + (size_t)ray * 3;
        const float* pd = p.rays_d + (size_t)ray * 3;
        r.ox = po[0]; r.oy = po[1]; r.oz = po[2];
        r.dx = pd[0]; r.dy = pd[1]; r.dz = pd[2];
        return r;
    }
    const int pose = ray / p.rays_per_pose;
    const int pix = p.pix_begin + (ray - pose * p.rays_per_pose);
    const int jrow = pix / p.W;
    const int icol = pix - jrow * p.W;
    float c0, c1, c2, c3, c4, c5, c6, c7, c8, c9, c10, c11;
    if (p.c2w != nullptr) {
        const float* c = p.c2w + (size_t)pose * 12;
        c0 = c[0]; c1 = c[1]; c2 = c[2]; c3 = c[3]; c4 = c[4]; c5 = c[5];
        c6 = c[6]; c7 = c[7]; c8 = c[8]; c9 = c[9]; c10 = c[10]; c11 = c[11];
    } else {
        c0 = p.c2w_host[0]; c1 = p.c2w_host[1]; c2 = p.c2w_host[2]; c3 = p.c2w_host[3];
        c4 = p.c2w_host[4]; c5 = p.c2w_host[5]; c6 = p.c2w_host[6]; c7 = p.c2w_host[7];
        c8 = p.c2w_host[8]; c9 = p.c2w_host[9]; c10 = p.c2w_host[10]; c11 = p.c2w_host[11];
    }
    const float dx = __fdiv_rn((float)icol - p.half_w, p.focal);
    const float dy = -__fdiv_rn((float)jrow - p.half_h, p.focal);
    // torch.sum(dirs[..., None, :] * c2w[:3,:3], -1): products rounded, then summed left to right
    r.dx = __fadd_rn(__fadd_rn(__fmul_rn(dx, c0), __fmul_rn(dy, c1)), __fmul_rn(-1.0f, c2));
    r.dy = __fadd_rn(__fadd_rn(__fmul_rn(dx, c4), __fmul_rn(dy, c5)), __fmul_rn(-1.0f, c6));
    r.dz = __fadd_rn(__fadd_rn(__fmul_rn(dx, c8), __fmul_rn(dy, c9)), __fmul_rn(-1.0f, c10));
    r.ox = c3; r.oy = c7; r.oz = c11;
    return r;
}

__device__ __forceinline__ float sample_pt(float o, float d, float z) {
    return __fadd_rn(o, __fmul_rn(d, z));  // rays_o + rays_d * z  (two roundings)
}

// ------------------------------------------------------------------------------------
// stand-alone K1+K2
// ------------------------------------------------------------------------------------
__global__ void r2l_sample_embed_kernel(R2LParams p, float* __restrict__ pts_out,
                                        float* __restrict__ emb_out) {
    // one thread per (ray, coordinate): 21 contiguous outputs
    long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)p.n_rays * R2L_NCOORD;
    if (gid >= total) return;
    int ray = (int)(gid / R2L_NCOORD);
    int c = (int)(gid - (long long)ray * R2L_NCOORD);
    int s = c / 3, kk = c - 3 * s;
    const Ray6 r6 = make_ray(p, ray);
    float x = sample_pt(r6.o(kk), r6.d(kk), p.z[s]);
    if (pts_out) pts_out[gid] = x;
    if (emb_out) {
        float* e = emb_out + gid * R2L_EMBED;
        Rev r = to_rev(x);
        float pw = 1.0f;
#pragma unroll
        for (int l = 0; l < R2L_L; ++l) {
            e[l] = trig_pow2(r, pw, false);
            e[R2L_L + l] = trig_pow2(r, pw, true);
            pw *= 2.0f;
        }
        e[2 * R2L_L] = x;
    }
}

__global__ void r2l_embed_kernel(const float* __restrict__ x_in, long long total, int L,
                                 float* __restrict__ emb_out) {
    long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    float x = x_in[gid];
    float* e = emb_out + gid * (2 * L + 1);
    Rev r = to_rev(x);
    float pw = 1.0f;
    for (int l = 0; l < L; ++l) {
        e[l] = trig_pow2(r, pw, false);
        e[L + l] = trig_pow2(r, pw, true);
        pw *= 2.0f;
    }
    e[2 * L] = x;
}

// ------------------------------------------------------------------------------------
// fused persistent kernel
// ------------------------------------------------------------------------------------
// epilogue of accumulator register r of row tile u (one column tile); activations are kept
// multiplied by act_scale (`inv` = act_scale / layer scale):
// SECOND = false: out = inact(acc/scale)            (ResMLP body.0 + inact)
// SECOND = true : x = outact(x + acc/scale); out = x (ResMLP body.2 + residual [+ outact])
// slope: the activation as act(v) = max(v, slope v): 0 relu (the exact fmaxf(v, 0) form), 0.01 LeakyReLU, 1 none
// GA = false: the README's network (relu / relu / none), the code of rounds 1-3 exactly; GA = true: slopes from the parameters
// (a second instantiation of the kernel: as run-time selects in the one kernel they cost the relu network 21 %)
template <bool GA, bool SECOND>
__device__ __forceinline__ float r2l_act(float v, float slope) {
    if (!GA) return SECOND ? v : fmaxf(v, 0.0f);
    return slope == 0.0f ? fmaxf(v, 0.0f) : (slope == 1.0f ? v : fmaxf(v, slope * v));
}
template <int NP, bool SECOND, bool GA>
__device__ __forceinline__ void epi_reg(const f32x4& acc, float inv, float neg1, f32x4& xu, f16x8& nh, f16x8& nl, int u,
                                        int r, float slope, float resid) {
    // r = pair index (0, 1): registers 2r, 2r+1 -> dword 2(u&1) + r of the fragments
    float v[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (!SECOND) {
            v[k] = r2l_act<GA, false>(acc[2 * r + k] * inv, slope);
        } else {
            // GA: resid = 0 drops the block's residual (trial.body_arch = mlp: two plain layers per "block")
            v[k] = r2l_act<GA, true>(fmaf(acc[2 * r + k], inv, GA ? resid * xu[2 * r + k] : xu[2 * r + k]), slope);
            xu[2 * r + k] = v[k];
        }
    }
    split_store2<NP>(v[0], v[1], nh, nl, 2 * (u & 1) + r, neg1);
}

// One row tile (16 output features) of a body layer: 8 k-steps x 2 column tiles.  A
// fragments are read one step ahead (the chunk's last step prefetches fragment 0 of the next
// chunk, certified by this chunk's ring_mid).  The 8 accumulator values of the previous row
// tile get their VALU epilogue interleaved, one register pair every second k-step.
template <int NP, bool SECOND, bool HAVE_PREV, bool GA>
__device__ __forceinline__ void body_rtile(Ring<NP>& R, int upos, const f16x8 (&Bh)[8][2], const f16x8 (&Bl)[8][2],
                                           f16x8 (&Nh)[8][2], f16x8 (&Nl)[8][2], f32x4 (&acc)[2],
                                           const f32x4 (&prev)[2], float inv, float neg1, f32x4 (&xprev)[2], int uprev, int q,
                                           float slope, float resid) {
    const uint32_t slot = R.use_off;
    const uint32_t lane_base = slot + R.lane * 16;
    const uint32_t next_base = ring_next_off<NP>(slot) + R.lane * 16;
    acc[0] = acc_init<NP>(slot, 16 * upos, q);
    acc[1] = acc[0];
#pragma unroll
    for (int s = 0; s < R2L_KSTEPS; ++s) {
        const int f = upos * R2L_KSTEPS + s;
        ring_step<NP>(R, f);  // f == 8: rendezvous + refill burst
        AFrag<NP> nxt = (f + 1 < R2L_FRAGS) ? read_frag<NP>(lane_base, f + 1) : read_frag<NP>(next_base, 0);
        acc[0] = mfma_step<NP>(R.pre, Bh[s][0], Bl[s][0], acc[0]);
        acc[1] = mfma_step<NP>(R.pre, Bh[s][1], Bl[s][1], acc[1]);
        if (HAVE_PREV && (s & 1))
            epi_reg<NP, SECOND, GA>(prev[s >> 2], inv, neg1, xprev[s >> 2], Nh[uprev >> 1][s >> 2], Nl[uprev >> 1][s >> 2], uprev,
                                (s >> 1) & 1, slope, resid);
        R.pre = nxt;
    }
    if (upos == 1) ring_next<NP>(R);
}

// One body Linear(256,256): in = (Bh,Bl) fragments, out fragments -> (Nh,Nl).
template <int NP, bool SECOND, bool GA>
__device__ __forceinline__ void body_layer(Ring<NP>& R, const f16x8 (&Bh)[8][2], const f16x8 (&Bl)[8][2],
                                           f16x8 (&Nh)[8][2], f16x8 (&Nl)[8][2], f32x4 (&x)[16][2],
                                           float act_scale, float neg1, int q, float slope, float resid) {
    const float inv = aux_inv_scale<NP>(R.use_off) * act_scale;  // same for the 8 chunks of a layer
    f32x4 acc[2], prev[2];
    body_rtile<NP, SECOND, false, GA>(R, 0, Bh, Bl, Nh, Nl, acc, prev, inv, neg1, x[0], 0, q, slope, resid);
#pragma unroll
    for (int u = 1; u < R2L_RTILES; ++u) {
        prev[0] = acc[0];
        prev[1] = acc[1];
        body_rtile<NP, SECOND, true, GA>(R, u & 1, Bh, Bl, Nh, Nl, acc, prev, inv, neg1, x[u - 1], u - 1, q, slope, resid);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        epi_reg<NP, SECOND, GA>(acc[i >> 1], inv, neg1, x[R2L_RTILES - 1][i >> 1], Nh[(R2L_RTILES - 1) >> 1][i >> 1],
                            Nl[(R2L_RTILES - 1) >> 1][i >> 1], R2L_RTILES - 1, i & 1, slope, resid);
}

// one head k-step (one chunk): 16 row tiles against the generated B fragments of both column
// tiles.  GEN is a statement using `i` (0..15), expanded between the MFMA groups: it produces
// element (i>>3, i&7) of the NEXT k-step's fragments, so the embedding VALU work runs under this
// step's MFMAs.  (A macro, not a functor: arrays captured by a lambda ended up in scratch.)
#define R2L_HEAD_STEP(BH, BL, ...)                                                                      \
    do {                                                                                               \
        const uint32_t lane_base_ = R.use_off + R.lane * 16;                                           \
        const uint32_t next_base_ = ring_next_off<NP>(R.use_off) + R.lane * 16;                        \
        _Pragma("unroll") for (int i = 0; i < R2L_RTILES; ++i) {                                       \
            ring_step<NP>(R, i);                                                                       \
            AFrag<NP> nxt_ = (i + 1 < R2L_FRAGS) ? read_frag<NP>(lane_base_, i + 1) : read_frag<NP>(next_base_, 0); \
            x[i][0] = mfma_step<NP>(R.pre, (BH)[0], (BL)[0], x[i][0]);                                 \
            x[i][1] = mfma_step<NP>(R.pre, (BH)[1], (BL)[1], x[i][1]);                                 \
            R.pre = nxt_;                                                                              \
            __VA_ARGS__;                                                                               \
        }                                                                                              \
        ring_next<NP>(R);                                                                              \
    } while (0)

__device__ __forceinline__ float sel4(int q, float a, float b, float c, float d) {
    return (q & 2) ? ((q & 1) ? d : c) : ((q & 1) ? b : a);
}

template <int NP, bool GA>
__global__ __launch_bounds__(256, 1) void r2l_resmlp_kernel(R2LParams p) {
    typedef KCfg<NP> C;
    Ring<NP> R;
    R.wimg = p.wimg;
    R.cpt = p.chunks_per_tile;
    R.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    R.lane = threadIdx.x & 63;
    R.issue_pos = 0;
    R.issue_off = 0;
    R.use_off = 0;
    const int lane = R.lane;
    const int q = lane >> 4;
    const float act_scale = p.act_scale;
    const float neg1 = p.neg1;  // -1.0f, opaque to the compiler (r2l_device.h pack_lo)
    // read-only, wave-uniform: constant address space => scalar loads (s_load), no vmcnt traffic
    const __attribute__((address_space(4))) float* zc = (const __attribute__((address_space(4))) float*)p.z;

    // prologue: D chunks in flight, chunk 0 certified, its fragment 0 in registers
#pragma unroll
    for (int i = 0; i < C::D; ++i) ring_issue<NP>(R);
    R2L_WAIT_VMCNT(C::WAIT_PRO);
    __builtin_amdgcn_s_barrier();
    R.pre = read_frag<NP>(lane * 16, 0);

    f32x4 x[16][2];
    f16x8 Bh[8][2], Bl[8][2], Nh[8][2], Nl[8][2];   // fp16 fragments (Bl / Nl: lo parts)

    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        // a lane serves ray (lane & 15) of both column tiles (p.tile_begin: this launch is a slice of the call)
        Ray6 rr[2];
        int ray[2];
        bool valid[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ray_raw = (p.tile_begin + tile) * R2L_TILE_RAYS + R.wave * R2L_RAYS_PER_WAVE + c * 16 + (lane & 15);
            valid[c] = ray_raw < p.n_rays;
            ray[c] = valid[c] ? ray_raw : p.n_rays - 1;
            rr[c] = make_ray(p, ray[c]);
        }

        // ---------------- head: Linear(1008,256) + ReLU, k outer / row tile inner ------------
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            x[u][0] = acc_init<NP>(R.use_off, 16 * u, q);
            x[u][1] = x[u][0];
        }
        // phase 1: k-steps 0..23: coordinates 2s, 2s+1 (by q>>1), frequencies 0..7, sin|cos by q&1.
        // Three steps (= two samples) per iteration; inside it the fragments of step u3+1 are
        // generated under the MFMAs of step u3.
        const bool is_cos1 = (q & 1) != 0;
        for (int sp = 0; sp < 8; ++sp) {
            const float z0 = zc[2 * sp], z1 = zc[2 * sp + 1];
            float xsel[3][2];
#pragma unroll
            for (int u3 = 0; u3 < 3; ++u3) {
                const int ca = 2 * u3, cb = 2 * u3 + 1;  // coordinates inside this pair of samples
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float xa = sample_pt(rr[c].o(ca % 3), rr[c].d(ca % 3), ca < 3 ? z0 : z1);
                    const float xb = sample_pt(rr[c].o(cb % 3), rr[c].d(cb % 3), cb < 3 ? z0 : z1);
                    xsel[u3][c] = (q & 2) ? xb : xa;
                }
            }
            f16x8 bh[2][2], bl[2][2];  // [buffer][column tile]
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const Rev r = to_rev(xsel[0][c]);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    split_store<NP>(trig_pow2(r, (float)(1 << j), is_cos1) * act_scale, bh[0][c], bl[0][c], j);
            }
#pragma unroll
            for (int u3 = 0; u3 < 3; ++u3) {
                Rev rn[2];
                R2L_HEAD_STEP(bh[u3 & 1], bl[u3 & 1], {
                    if (u3 < 2) {
                        const int c = i >> 3, j = i & 7;
                        if (j == 0) rn[c] = to_rev(xsel[u3 < 2 ? u3 + 1 : 0][c]);
                        split_store<NP>(trig_pow2(rn[c], (float)(1 << j), is_cos1) * act_scale, bh[(u3 & 1) ^ 1][c],
                                        bl[(u3 & 1) ^ 1][c], j);
                    }
                });
            }
        }
        // phase 2: k-steps 24..29: coordinates 8(s-24) + 2q + (j>>2), frequencies 8, 9; same pipelining
        for (int it = 0; it < 2; ++it) {
            Rev r01[3][2][2];  // [step][column tile][coordinate of the pair]
#pragma unroll
            for (int u3 = 0; u3 < 3; ++u3) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    float pt[8];
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        const int cc = 8 * u3 + m;  // coordinate inside this group of 24
                        pt[m] = sample_pt(rr[c].o(cc % 3), rr[c].d(cc % 3), zc[8 * it + cc / 3]);
                    }
                    r01[u3][c][0] = to_rev(sel4(q, pt[0], pt[2], pt[4], pt[6]));
                    r01[u3][c][1] = to_rev(sel4(q, pt[1], pt[3], pt[5], pt[7]));
                }
            }
            f16x8 bh[2][2], bl[2][2];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    split_store<NP>(trig_pow2(r01[0][c][j >> 2], (j & 1) ? 512.0f : 256.0f, ((j >> 1) & 1) != 0) * act_scale,
                                    bh[0][c], bl[0][c], j);
#pragma unroll
            for (int u3 = 0; u3 < 3; ++u3) {
                R2L_HEAD_STEP(bh[u3 & 1], bl[u3 & 1], {
                    if (u3 < 2) {
                        const int c = i >> 3, j = i & 7;
                        split_store<NP>(trig_pow2(r01[u3 < 2 ? u3 + 1 : 0][c][j >> 2], (j & 1) ? 512.0f : 256.0f,
                                                  ((j >> 1) & 1) != 0) * act_scale,
                                        bh[(u3 & 1) ^ 1][c], bl[(u3 & 1) ^ 1][c], j);
                    }
                });
            }
        }
        // phase 3: k-steps 30, 31: identity of coordinate 8q + j (+32; quarters 2, 3 of step 31 pad)
        float inv_head = 0.f;
#pragma unroll
        for (int s3 = 0; s3 < 2; ++s3) {
            f16x8 bh[2], bl[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float v;
                    if (s3 == 0) {
                        const int c0 = j, c1 = 8 + j, c2 = 16 + j, c3 = 24 + j;
                        v = sel4(q, sample_pt(rr[c].o(c0 % 3), rr[c].d(c0 % 3), zc[c0 / 3]),
                                 sample_pt(rr[c].o(c1 % 3), rr[c].d(c1 % 3), zc[c1 / 3]),
                                 sample_pt(rr[c].o(c2 % 3), rr[c].d(c2 % 3), zc[c2 / 3]),
                                 sample_pt(rr[c].o(c3 % 3), rr[c].d(c3 % 3), zc[c3 / 3]));
                    } else {
                        const int c0 = 32 + j, c1 = 40 + j;
                        v = sel4(q, sample_pt(rr[c].o(c0 % 3), rr[c].d(c0 % 3), zc[c0 / 3]),
                                 sample_pt(rr[c].o(c1 % 3), rr[c].d(c1 % 3), zc[c1 / 3]), 0.0f, 0.0f);
                    }
                    split_store<NP>(v * act_scale, bh[c], bl[c], j);
                }
            }
            if (s3 == 1) inv_head = aux_inv_scale<NP>(R.use_off) * act_scale;  // chunk 31's aux
            R2L_HEAD_STEP(bh, bl, {});
        }

        // head epilogue: h0 = act(acc/scale) (scaled domain); keep a copy for the global skip
        float* scr = p.scratch + ((size_t)(blockIdx.x * R2L_WAVES + R.wave) * 32) * 256 + lane * 4;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = r2l_act<GA, false>(x[u][c][r] * inv_head, p.act_head);
                    x[u][c][r] = v;
                    split_store<NP>(v, Bh[u >> 1][c], Bl[u >> 1][c], 4 * (u & 1) + r);
                }
                if (p.use_residual) *reinterpret_cast<f32x4*>(scr + (u * 2 + c) * 256) = x[u][c];
            }
        }

        // ---------------- body: n_block x ResMLP ----------------------------------------
        for (int blk = 0; blk < p.n_block; ++blk) {
            body_layer<NP, false, GA>(R, Bh, Bl, Nh, Nl, x, act_scale, neg1, q, p.act_in, 1.0f);
            body_layer<NP, true, GA>(R, Nh, Nl, Bh, Bl, x, act_scale, neg1, q, p.act_out, p.block_resid);
        }

        // ---------------- global skip + tail: sigmoid(Linear(256,3)) -----------------------
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (p.use_residual) {
                    const f32x4 h0 = *reinterpret_cast<const f32x4*>(scr + (u * 2 + c) * 256);
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[u][c][r] += h0[r];      // element-wise on purpose: no packed-fp32 VALU (csrc/Makefile)
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) split_store<NP>(x[u][c][r], Bh[u >> 1][c], Bl[u >> 1][c], 4 * (u & 1) + r);
            }
        }
        {
            // tail chunk: row tile 0 only (fragments 0..7 real), then leave the chunk
            const uint32_t slot = R.use_off;
            const uint32_t lane_base = slot + lane * 16;
            const float inv = aux_inv_scale<NP>(slot);
            f32x4 acc[2];
            acc[0] = acc_init<NP>(slot, 0, q);
            acc[1] = acc[0];
#pragma unroll
            for (int s = 0; s < R2L_KSTEPS; ++s) {
                AFrag<NP> nxt = read_frag<NP>(lane_base, s + 1);  // s = 7 reads the unused fragment 8
                acc[0] = mfma_step<NP>(R.pre, Bh[s][0], Bl[s][0], acc[0]);
                acc[1] = mfma_step<NP>(R.pre, Bh[s][1], Bl[s][1], acc[1]);
                R.pre = nxt;
            }
            ring_mid<NP>(R);
            R.pre = read_frag<NP>(ring_next_off<NP>(slot) + lane * 16, 0);
            ring_next<NP>(R);
            if (q == 0) {  // rows 0..2 of the tile = rgb, in the quarter-0 lanes
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if (valid[c]) {
                        float* out = p.rgb + (size_t)ray[c] * 3;
#pragma unroll
                        for (int k = 0; k < 3; ++k) out[k] = 1.0f / (1.0f + expf(-(acc[c][k] * inv)));
                    }
                }
            }
        }
    }
    // the ring always runs D chunks ahead: drain before the LDS is released
    R2L_WAIT_VMCNT(0);
}

// ------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------
template <int NP, bool GA>
static hipError_t launch_resmlp_ga(const R2LParams& p, int grid, hipStream_t stream) {
    // the > 64 KiB dynamic-LDS opt-in is per device: a process may drive several GPUs
    static std::atomic<bool> attr_set[64];  // zero-initialised; the opt-in call itself is idempotent
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&r2l_resmlp_kernel<NP, GA>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, KCfg<NP>::LDS);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL((r2l_resmlp_kernel<NP, GA>), dim3(grid), dim3(256), KCfg<NP>::LDS, stream, p);
    return hipGetLastError();
}
template <int NP>
static hipError_t launch_resmlp(const R2LParams& p, int grid, hipStream_t stream) {
    // relu / relu / none: the specialised instantiation; any other slopes (r2l_set_activations): the general one
    if (p.act_head == 0.0f && p.act_in == 0.0f && p.act_out == 1.0f && p.block_resid == 1.0f) return launch_resmlp_ga<NP, false>(p, grid, stream);
    return launch_resmlp_ga<NP, true>(p, grid, stream);
}

// ------------------------------------------------------------------------------------
// FP16_FP8 head layer: h0 = relu(W_h embedding + b_h) -> p.xbuf (register image of r2l_body_kernel).  HIP code makes the
// lane's camera ray; the layer itself, embedding included, is ONE inline-asm block per 128-ray tile generated by
// gen/head_gen.py (r2l_head_asm.inc): the 63 embedding values of point p + 1 are computed in registers under the MFMAs
// of point p (32x32 shapes, fp16 pass + two bf6 terms), k-outer into the 128 accumulator registers that start from the
// bias.  The block owns v0-v209, a0-a127, s40-s59, vcc.  The generator's emulator checks the stream against a float64
// evaluation of the layer (tests/test_head_gen_cpu.py).
// ------------------------------------------------------------------------------------
// X3: gen/head_gen.py --fmt f16 (r2l_headx_asm.inc): three fp16 passes per k-step, no bf6 terms (R2L_PREC_FP16X3_ASM).
template <bool X3>
__global__ __launch_bounds__(256, 1) void r2l_head_kernel(R2LParams p) {
    extern __shared__ __attribute__((aligned(16))) char r2l_head_lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    {   // resident table: bias x act_scale | E8M0 weight scales
        const uint4* src = reinterpret_cast<const uint4*>(p.wimg + (X3 ? R2L_HEADX_STREAM_BYTES : R2L_HEAD_STREAM_BYTES));
        uint4* dst = reinterpret_cast<uint4*>(r2l_head_lds + 4 * (X3 ? R2L_HEADX_CHUNK : 28672));
        if (threadIdx.x < R2L_HEAD_AUX_BYTES / 16) dst[threadIdx.x] = src[threadIdx.x];
    }
    __syncthreads();
    if constexpr (X3) {
        asm volatile(
#include "r2l_headx_pro_asm.inc"
            :
            : [wimg] "s"(p.wimg), [wave] "s"(wave)
            :
#include "r2l_headx_pro_clobbers.inc"
        );
    } else {
        asm volatile(
#include "r2l_head_pro_asm.inc"
            :
            : [wimg] "s"(p.wimg), [wave] "s"(wave)
            :
#include "r2l_head_pro_clobbers.inc"
        );
    }
    // read-only, wave-uniform: constant address space => scalar loads
    const __attribute__((address_space(4))) float* zc = (const __attribute__((address_space(4))) float*)p.z;
    const float z0 = zc[0], z1 = zc[1], z2 = zc[2], z3 = zc[3], z4 = zc[4], z5 = zc[5], z6 = zc[6], z7 = zc[7], z8 = zc[8],
                z9 = zc[9], z10 = zc[10], z11 = zc[11], z12 = zc[12], z13 = zc[13], z14 = zc[14], z15 = zc[15];
    // running maximum of h0 over this lane's rays of the launch: an in/out operand of the generated block, which names it v210
    register float hmax asm("v210") = 0.0f;
    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        // lane 32 h + r serves ray r of the wave (p.tile_begin: this launch is a slice of the call); rays past the end
        // repeat the last one (the tail launch masks their stores)
        const int ray_raw = (p.tile_begin + tile) * R2L_TILE_RAYS + wave * R2L_RAYS_PER_WAVE + (lane & 31);
        const Ray6 r = make_ray(p, ray_raw < p.n_rays ? ray_raw : p.n_rays - 1);
        float* xout = p.xbuf + ((size_t)(tile * R2L_WAVES + wave) * 32) * 256;
#define R2L_HEAD_BLOCK_OPERANDS                                                                                          \
            : [hmax] "+v"(hmax)                                                                                          \
            : [wimg] "s"(p.wimg), [wave] "s"(wave), [xout] "s"(xout), [o0] "v"(r.ox), [o1] "v"(r.oy), [o2] "v"(r.oz),     \
              [d0] "v"(r.dx), [d1] "v"(r.dy), [d2] "v"(r.dz), [z0] "s"(z0), [z1] "s"(z1), [z2] "s"(z2), [z3] "s"(z3),    \
              [z4] "s"(z4), [z5] "s"(z5), [z6] "s"(z6), [z7] "s"(z7), [z8] "s"(z8), [z9] "s"(z9), [z10] "s"(z10),        \
              [z11] "s"(z11), [z12] "s"(z12), [z13] "s"(z13), [z14] "s"(z14), [z15] "s"(z15)
        if constexpr (X3) {
            asm volatile(
#include "r2l_headx_asm.inc"
                R2L_HEAD_BLOCK_OPERANDS
                :
#include "r2l_headx_clobbers.inc"
            );
        } else {
            asm volatile(
#include "r2l_head_asm.inc"
                R2L_HEAD_BLOCK_OPERANDS
                :
#include "r2l_head_clobbers.inc"
            );
        }
#undef R2L_HEAD_BLOCK_OPERANDS
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last block's refill of the ring and its stores
    if (p.range != nullptr) {   // every ray of the launch: rays past the end repeat the last one, so they add nothing foreign
        float m = hmax;
#pragma unroll
        for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, __shfl_xor(m, k));
        if (lane == 0 && m > 0.0f) atomicMax(p.range, __float_as_uint(m));
    }
}

hipError_t r2l_launch_head(const R2LParams& p, int grid, hipStream_t stream, int x3) {
    static std::atomic<bool> attr_set[64];  // zero-initialised; the opt-in call itself is idempotent
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&r2l_head_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                R2L_HEAD_LDS);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&r2l_head_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    R2L_HEADX_LDS);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    if (x3) hipLaunchKernelGGL(r2l_head_kernel<true>, dim3(grid), dim3(256), R2L_HEADX_LDS, stream, p);
    else hipLaunchKernelGGL(r2l_head_kernel<false>, dim3(grid), dim3(256), R2L_HEAD_LDS, stream, p);
    return hipGetLastError();
}

hipError_t r2l_launch_resmlp(const R2LParams& p, int mode, int grid, hipStream_t stream) {
    return mode == R2L_PREC_FP16X3 ? launch_resmlp<2>(p, grid, stream) : launch_resmlp<1>(p, grid, stream);
}

hipError_t r2l_launch_sample_embed(const R2LParams& p, float* pts_out, float* emb_out,
                                   hipStream_t stream) {
    long long total = (long long)p.n_rays * R2L_NCOORD;
    int grid = (int)((total + 255) / 256);
    hipLaunchKernelGGL(r2l_sample_embed_kernel, dim3(grid), dim3(256), 0, stream, p, pts_out, emb_out);
    return hipGetLastError();
}

hipError_t r2l_launch_embed(const float* x, long long total, int L, float* emb_out,
                            hipStream_t stream) {
    int grid = (int)((total + 255) / 256);
    hipLaunchKernelGGL(r2l_embed_kernel, dim3(grid), dim3(256), 0, stream, x, total, L, emb_out);
    return hipGetLastError();
}
