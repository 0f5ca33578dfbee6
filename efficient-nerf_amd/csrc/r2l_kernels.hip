// R2L (neural light field) hot path for MI355X / gfx950, hand-written HIP.
//
//   r2l_resmlp_kernel<NP>  K1+K2+K3 fused: get_rays + 16-point sampling + sinusoidal
//                          embedding + 88-layer width-256 residual MLP + sigmoid, one
//                          persistent workgroup per CU, activations resident in registers,
//                          weights streamed global -> LDS ring (LDS-DMA) -> MFMA A operand.
//                          NP = 2: fp16 hi/lo split, 3 MFMA per k-step (fp32-grade result)
//                          NP = 1: single fp16 pass.
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

#include "r2l_common.h"
#include "r2l_kernels.h"

#include "r2l_device.h"

// camera ray of flat ray index `ray` (model/nerf_raybased.py:84-99).
__device__ __forceinline__ void make_ray(const R2LParams& p, int ray, float o[3], float d[3]) {
    if (p.rays_o != nullptr) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            o[k] = p.rays_o[(size_t)ray * 3 + k];
            d[k] = p.rays_d[(size_t)ray * 3 + k];
        }
        return;
    }
    int pose = ray / p.rays_per_pose;
    int pix = p.pix_begin + (ray - pose * p.rays_per_pose);
    int jrow = pix / p.W;
    int icol = pix - jrow * p.W;
    float c[12];
    if (p.c2w != nullptr) {
#pragma unroll
        for (int k = 0; k < 12; ++k) c[k] = p.c2w[(size_t)pose * 12 + k];
    } else {
#pragma unroll
        for (int k = 0; k < 12; ++k) c[k] = p.c2w_host[k];
    }
    float dx = __fdiv_rn((float)icol - p.half_w, p.focal);
    float dy = -__fdiv_rn((float)jrow - p.half_h, p.focal);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        // torch.sum(dirs[..., None, :] * c2w[:3,:3], -1): products rounded, then summed
        float s = __fadd_rn(__fmul_rn(dx, c[4 * k + 0]), __fmul_rn(dy, c[4 * k + 1]));
        d[k] = __fadd_rn(s, __fmul_rn(-1.0f, c[4 * k + 2]));
        o[k] = c[4 * k + 3];
    }
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
    float o[3], d[3];
    make_ray(p, ray, o, d);
    float ok = kk == 0 ? o[0] : (kk == 1 ? o[1] : o[2]);
    float dk = kk == 0 ? d[0] : (kk == 1 ? d[1] : d[2]);
    float x = sample_pt(ok, dk, p.z[s]);
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
// epilogue of one accumulator register of feature tile t:
// SECOND = false: out = relu(acc/scale)              (ResMLP body.0 + inact)
// SECOND = true : x = x + acc/scale; out = x         (ResMLP body.2 + residual)
template <int NP, bool SECOND>
__device__ __forceinline__ void epi_reg(const f32x16& acc, float inv, f32x16& xt, f16x8 (&Nh)[16],
                                        f16x8 (&Nl)[16], int t, int reg, float act_scale) {
#ifdef R2L_ABL_NOEPI  // ablation build: keep the accumulator live, skip the VALU epilogue
    asm volatile("" ::"v"(acc[reg]));
    return;
#endif
    float v;  // activation * act_scale (`inv` = act_scale / layer scale)
    if (!SECOND) {
        v = fmaxf(acc[reg] * inv, 0.0f);
    } else {
        v = fmaf(acc[reg], inv, xt[reg]);
        xt[reg] = v;
    }
    split_store<NP>(v, Nh[2 * t + (reg >> 3)], Nl[2 * t + (reg >> 3)], reg & 7);
}

// 16 k-steps of one feature tile (one chunk), A fragments prefetched one step ahead (the
// last step prefetches fragment 0 of the next chunk, certified by this chunk's ring_mid).
// While the MFMAs of tile t run, the VALU epilogue of tile t-1 (`prev`) is interleaved,
// one accumulator register per k-step.
template <int NP, bool SECOND, bool HAVE_PREV>
__device__ __forceinline__ f32x16 body_tile(Ring<NP>& R, const f16x8 (&Bh)[16], const f16x8 (&Bl)[16],
                                            f16x8 (&Nh)[16], f16x8 (&Nl)[16], const f32x16& prev,
                                            float inv, f32x16& xprev, int tprev, float act_scale,
                                            int h) {
    const uint32_t slot = R.use_off;
    const uint32_t lane_base = slot + R.lane * 16;
    const uint32_t next_base = ring_next_off<NP>(slot) + R.lane * 16;
    f32x16 acc = acc_init<NP>(slot, 0, h);
    AFrag<NP> cur = R.pre;
#pragma unroll
    for (int ks = 0; ks < R2L_KSTEPS; ++ks) {
        if (ks == R2L_KSTEPS / 2) ring_mid<NP>(R);
        AFrag<NP> nxt = (ks + 1 < R2L_KSTEPS) ? read_frag<NP>(lane_base, ks + 1) : read_frag<NP>(next_base, 0);
        acc = mfma_step<NP>(cur, Bh[ks], Bl[ks], acc);
        if (HAVE_PREV) epi_reg<NP, SECOND>(prev, inv, xprev, Nh, Nl, tprev, ks, act_scale);
        cur = nxt;
    }
    R.pre = cur;
    ring_next<NP>(R);
    return acc;
}

// One body Linear(256,256): in = (Bh,Bl) fragments, out fragments -> (Nh,Nl).
template <int NP, bool SECOND>
__device__ __forceinline__ void body_layer(Ring<NP>& R, const f16x8 (&Bh)[16], const f16x8 (&Bl)[16],
                                           f16x8 (&Nh)[16], f16x8 (&Nl)[16], f32x16 (&x)[8],
                                           float act_scale, int h) {
    // activations (x, h0, B fragments) live multiplied by act_scale: `inv` only removes the weight scale
    const float inv = aux_inv_scale<NP>(R.use_off) * act_scale;  // same for the 8 chunks of a layer
    f32x16 prev = body_tile<NP, SECOND, false>(R, Bh, Bl, Nh, Nl, x[0], inv, x[0], 0, act_scale, h);
#pragma unroll
    for (int t = 1; t < R2L_NTILE; ++t)
        prev = body_tile<NP, SECOND, true>(R, Bh, Bl, Nh, Nl, prev, inv, x[t - 1], t - 1, act_scale, h);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg)
        epi_reg<NP, SECOND>(prev, inv, x[R2L_NTILE - 1], Nh, Nl, R2L_NTILE - 1, reg, act_scale);
}

// one head k-step: 8 feature tiles against one generated B fragment (fragments ksl*8 + t)
template <int NP>
__device__ __forceinline__ void head_kstep(Ring<NP>& R, int ksl, const f16x8& bh, const f16x8& bl,
                                           f32x16 (&x)[8]) {
    const uint32_t lane_base = R.use_off + R.lane * 16;
    const uint32_t next_base = ring_next_off<NP>(R.use_off) + R.lane * 16;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int f = ksl * 8 + t;
        AFrag<NP> nxt = (f + 1 < R2L_FRAGS) ? read_frag<NP>(lane_base, f + 1) : read_frag<NP>(next_base, 0);
        x[t] = mfma_step<NP>(R.pre, bh, bl, x[t]);
        R.pre = nxt;
    }
    if (ksl == 0) ring_mid<NP>(R); else ring_next<NP>(R);
}

template <int NP>
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
    const int h = lane >> 5;
    const float act_scale = p.act_scale;
    const bool is_cos = h != 0;

    // prologue: D chunks in flight, chunk 0 certified, its fragment 0 in registers
#pragma unroll
    for (int i = 0; i < C::D; ++i) ring_issue<NP>(R);
    R2L_WAIT_VMCNT(C::WAIT_PRO);
    __builtin_amdgcn_s_barrier();
    R.pre = read_frag<NP>(lane * 16, 0);

    f32x16 x[8];
    f16x8 Bh[16], Bl[16], Nh[16], Nl[16];

    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        const int ray_raw = tile * R2L_TILE_RAYS + R.wave * R2L_RAYS_PER_WAVE + (lane & 31);
        const bool valid = ray_raw < p.n_rays;
        const int ray = valid ? ray_raw : p.n_rays - 1;
        float o[3], d[3];
        make_ray(p, ray, o, d);

        // ---------------- head: Linear(1008,256) + ReLU, k outer / feature tile inner -------
#pragma unroll
        for (int t = 0; t < 8; ++t) x[t] = acc_init<NP>(R.use_off, 32 * t, h);

        // phase 1: k-steps 0..47, one coordinate each, frequencies 0..7 (sin | cos by half)
        for (int sp = 0; sp < 8; ++sp) {
            const float z0 = p.z[2 * sp], z1 = p.z[2 * sp + 1];
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const float xc = sample_pt(o[u % 3], d[u % 3], u < 3 ? z0 : z1);
                const Rev r = to_rev(xc);
                f16x8 bh, bl;
                float pw = 1.0f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    split_store<NP>(trig_pow2(r, pw, is_cos) * act_scale, bh, bl, j);
                    pw *= 2.0f;
                }
                head_kstep<NP>(R, u & 1, bh, bl, x);
            }
        }
        // phase 2: k-steps 48..59, four coordinates each, frequencies 8, 9
        for (int it = 0; it < 2; ++it) {
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                f16x8 bh, bl;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int cc = 4 * u + m;  // coordinate within this group of 24
                    const float xc = sample_pt(o[cc % 3], d[cc % 3], p.z[8 * it + cc / 3]);
                    const Rev r = to_rev(xc);
                    split_store<NP>(trig_pow2(r, 256.0f, is_cos) * act_scale, bh, bl, 2 * m);
                    split_store<NP>(trig_pow2(r, 512.0f, is_cos) * act_scale, bh, bl, 2 * m + 1);
                }
                head_kstep<NP>(R, u & 1, bh, bl, x);
            }
        }
        // phase 3: k-steps 60..62 identity (coordinate 16*(ks-60) + 8h + j), k-step 63 = pad
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            f16x8 bh, bl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c0 = 16 * u + j, c1 = c0 + 8;
                const float x0 = sample_pt(o[c0 % 3], d[c0 % 3], p.z[c0 / 3]);
                const float x1 = sample_pt(o[c1 % 3], d[c1 % 3], p.z[c1 / 3]);
                split_store<NP>((h ? x1 : x0) * act_scale, bh, bl, j);
            }
            head_kstep<NP>(R, u & 1, bh, bl, x);
        }
        // pad k-step 63: no MFMAs; leave the chunk with the next chunk's fragment 0 prefetched
        const float inv_head = aux_inv_scale<NP>(R.use_off) * act_scale;  // h0, x: scaled domain
        R.pre = read_frag<NP>(ring_next_off<NP>(R.use_off) + lane * 16, 0);
        ring_next<NP>(R);

        // head epilogue: h0 = relu(acc/scale); keep a copy for the global skip
        float* scr = p.scratch + ((size_t)(blockIdx.x * R2L_WAVES + R.wave) * 32) * 256 + lane * 4;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                float v = fmaxf(x[t][reg] * inv_head, 0.0f);
                x[t][reg] = v;
                split_store<NP>(v, Bh[2 * t + (reg >> 3)], Bl[2 * t + (reg >> 3)], reg & 7);
            }
            if (p.use_residual) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v4 = {x[t][4 * g], x[t][4 * g + 1], x[t][4 * g + 2], x[t][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(scr + (t * 4 + g) * 256) = v4;
                }
            }
        }

        // ---------------- body: n_block x ResMLP ----------------------------------------
        for (int blk = 0; blk < p.n_block; ++blk) {
#ifdef R2L_ABL_NOEPI
            body_layer<NP, false>(R, Bh, Bl, Bh, Bl, x, act_scale, h);
            body_layer<NP, true>(R, Bh, Bl, Bh, Bl, x, act_scale, h);
#else
            body_layer<NP, false>(R, Bh, Bl, Nh, Nl, x, act_scale, h);
            body_layer<NP, true>(R, Nh, Nl, Bh, Bl, x, act_scale, h);
#endif
        }

        // ---------------- global skip + tail: sigmoid(Linear(256,3)) -----------------------
        if (p.use_residual) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v4 = *reinterpret_cast<const f32x4*>(scr + (t * 4 + g) * 256);
                    x[t][4 * g + 0] += v4[0];
                    x[t][4 * g + 1] += v4[1];
                    x[t][4 * g + 2] += v4[2];
                    x[t][4 * g + 3] += v4[3];
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                split_store<NP>(x[t][reg], Bh[2 * t + (reg >> 3)], Bl[2 * t + (reg >> 3)], reg & 7);
        {
            const float inv = aux_inv_scale<NP>(R.use_off);
            f32x16 acc = body_tile<NP, false, false>(R, Bh, Bl, Nh, Nl, x[0], inv, x[0], 0, act_scale, h);
            if (valid && h == 0) {
                float* out = p.rgb + (size_t)ray * 3;
#pragma unroll
                for (int k = 0; k < 3; ++k) out[k] = 1.0f / (1.0f + expf(-(acc[k] * inv)));
            }
        }
    }
    // the ring always runs D chunks ahead: drain before the LDS is released
    R2L_WAIT_VMCNT(0);
}

// ------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------
template <int NP>
static hipError_t launch_resmlp(const R2LParams& p, int grid, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&r2l_resmlp_kernel<NP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, KCfg<NP>::LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(r2l_resmlp_kernel<NP>, dim3(grid), dim3(256), KCfg<NP>::LDS, stream, p);
    return hipGetLastError();
}

hipError_t r2l_launch_resmlp(const R2LParams& p, int np, int grid, hipStream_t stream) {
    return np == 2 ? launch_resmlp<2>(p, grid, stream) : launch_resmlp<1>(p, grid, stream);
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
