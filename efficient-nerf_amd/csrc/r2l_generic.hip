// Generic fp32 layer path: the NeRF_v3_2 / NeRF variants the reference's constructors accept but the fused kernels are not
// built for (netwidth != 256, --layerwise_netwidths, trial.n_learnable != 2, n_sample_per_ray != 16, multires != 10, odd mlp
// depths; teacher netdepth / netwidth other than 8 x 256).  One launch per nn.Linear, activations through HBM:
//
//   r2l_linear_forward   y = post + act((x W^T + b) * res_scale + res)         nn.Linear + get_activation + ResMLP's residual
//                        (model/nerf_raybased.py:443-465, 468-476, 497-544; NeRF.forward :377-401)
//   nerf_embed           [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...]          Embedder.embed (utils/run_nerf_raybased_helpers.py:24-56)
//   r2l_sample_points    pts[r, 3 s + k] = o[r, k] + d[r, k] z[s]               PointSampler.sample_test / sample_train
//                        (model/nerf_raybased.py:100-102, 114-126), also main.py:701 with per-ray z
//
// Arithmetic: fp32 products, fp32 accumulate on the fp32 MFMA (v_mfma_f32_32x32x2_f32: 256 FLOP/clk/CU, 1/16 of the fp16
// rate) -- the reference's own precision, nothing to calibrate; the only difference from F.linear is the summation order
// (~2e-7 measured on ten network variants).  This is the path for the long tail of shapes, not the headline: the README's W256D88
// runs at 5.5e6 rays/s here (65 TFLOP/s, 0.41 of the fp32 MFMA peak; a 128 x 128 tile measures the same), 1/11 of the fused fp16_fp8
// kernels (profiles/r04_generic_time.txt).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/r2l_hip.h"
#include "r2l_host_util.h"

#include "r2l_device.h"      // to_rev / trig_pow2: sin / cos of the exactly scaled argument

struct r2l_linear {
    int in_dim, out_dim;
    float* w_dev;     // [out_dim][in_dim], the state_dict's layout
    float* b_dev;     // [out_dim] (zeros when the layer has no bias)
};

#define GT_M 128      // rays per workgroup (4 waves x 32)
#define GT_N 64       // output features per workgroup (2 MFMA column tiles per wave)
#define GT_K 32       // k per LDS stage

__device__ __forceinline__ float r2l_generic_act(float v, int act) {
    switch (act) {
        case R2L_ACT_RELU: return fmaxf(v, 0.0f);
        case R2L_ACT_LRELU: return v > 0.0f ? v : 0.01f * v;      // nn.LeakyReLU default slope
        case R2L_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
        default: return v;
    }
}

// D = A B + C on v_mfma_f32_32x32x2_f32: A lane l = x[ray l % 32][k l / 32], B lane l = W[out l % 32][k l / 32],
// D register r of lane l = y[ray 8 (r / 4) + 4 (l / 32) + r % 4][out l % 32]
__global__ __launch_bounds__(256) void r2l_linear_kernel(const float* __restrict__ x, long long ldx, int n, int in_dim,
                                                         const float* __restrict__ w, const float* __restrict__ b, int out_dim,
                                                         float* y, long long ldy, const float* res, long long ldr, float res_scale,
                                                         int act, const float* post, long long ldp) {
    __shared__ float xs[GT_M][GT_K + 1];
    __shared__ float ws[GT_N][GT_K + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long ray0 = (long long)blockIdx.x * GT_M;
    const int out0 = blockIdx.y * GT_N;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.0f;
    const int lc = tid & 31, lr = tid >> 5;      // loader: 8 rows of 32 consecutive k per pass
    // a stage's 24 loads per thread are issued together into registers (out-of-range elements read a valid address and are
    // zeroed: no branch between the loads), written to LDS behind the barrier, and the NEXT stage's loads are in flight while
    // this stage's MFMAs run
    // (the zeroing happens when the registers are written to LDS, so nothing waits for a load before the MFMAs)
    float xv[GT_M / 8], wv[GT_N / 8];
    unsigned okx = 0, okw = 0;
    auto load_stage = [&](int k0) {
        const int k = k0 + lc;
        const bool kin = k < in_dim;
        okx = okw = 0;
#pragma unroll
        for (int i = 0; i < GT_M / 8; ++i) {
            const long long ray = ray0 + lr + 8 * i;
            const bool ok = kin && ray < n;
            xv[i] = x[ok ? ray * ldx + k : 0];
            okx |= (ok ? 1u : 0u) << i;
        }
#pragma unroll
        for (int i = 0; i < GT_N / 8; ++i) {
            const int o = out0 + lr + 8 * i;
            const bool ok = kin && o < out_dim;
            wv[i] = w[ok ? (long long)o * in_dim + k : 0];
            okw |= (ok ? 1u : 0u) << i;
        }
    };
    load_stage(0);
    for (int k0 = 0; k0 < in_dim; k0 += GT_K) {
#pragma unroll
        for (int i = 0; i < GT_M / 8; ++i) xs[lr + 8 * i][lc] = ((okx >> i) & 1) ? xv[i] : 0.0f;
#pragma unroll
        for (int i = 0; i < GT_N / 8; ++i) ws[lr + 8 * i][lc] = ((okw >> i) & 1) ? wv[i] : 0.0f;
        __syncthreads();
        if (k0 + GT_K < in_dim) load_stage(k0 + GT_K);
#pragma unroll
        for (int kk = 0; kk < GT_K; kk += 2) {
            const int kq = kk + (lane >> 5);
            const float a = xs[wave * 32 + (lane & 31)][kq];
            const float b0 = ws[lane & 31][kq], b1 = ws[32 + (lane & 31)][kq];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int o = out0 + 32 * t + (lane & 31);
        if (o >= out_dim) continue;
        const float bias = b[o];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long ray = ray0 + wave * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
            if (ray >= n) continue;
            float v = (t ? acc1[r] : acc0[r]) + bias;
            if (res) v = v * res_scale + res[ray * ldr + o];          // ResMLP: body(x).mul(res_scale) + x, then outact
            v = r2l_generic_act(v, act);
            if (post) v = v + post[ray * ldp + o];                    // NeRF_v3_2.forward's global skip: body(x) + x
            y[ray * ldy + o] = v;
        }
    }
}

__global__ void r2l_sample_points_kernel(const float* __restrict__ ro, const float* __restrict__ rd, long long n,
                                         const float* __restrict__ z, int n_sample, int z_per_ray, float* __restrict__ pts) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n * n_sample) return;
    const long long r = gid / n_sample;
    const int s = (int)(gid - r * n_sample);
    const float zz = z_per_ray ? z[gid] : z[s];
#pragma unroll
    for (int k = 0; k < 3; ++k)      // rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]: one rounding per op
        pts[gid * 3 + k] = __fadd_rn(ro[r * 3 + k], __fmul_rn(rd[r * 3 + k], zz));
}

// Embedder.embed with include_input, log_sampling (utils/run_nerf_raybased_helpers.py:30-56; get_embedder :59-74): column block 0 is
// x, then per frequency 2^l a block of sin and a block of cos, each `dim` wide
__global__ void nerf_embed_kernel(const float* __restrict__ x_in, long long ldi, long long n, int dim, int L, float* __restrict__ out,
                                  long long ldo) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n * dim) return;
    const long long r = gid / dim;
    const int c = (int)(gid - r * dim);
    const float x = x_in[r * ldi + c];
    float* e = out + r * ldo + c;
    e[0] = x;
    const Rev rv = to_rev(x);
    float pw = 1.0f;
    for (int l = 0; l < L; ++l) {
        e[(long long)dim * (1 + 2 * l)] = trig_pow2(rv, pw, false);
        e[(long long)dim * (2 + 2 * l)] = trig_pow2(rv, pw, true);
        pw *= 2.0f;
    }
}

extern "C" {

int nerf_embed(const float* x_dev, long long ldi, int n, int dim, int multires, float* out_dev, long long ldo, void* stream) {
    if (!x_dev || !out_dev || n < 0 || dim <= 0 || multires < 0 || multires > 16 || ldi < dim || ldo < (long long)dim * (2 * multires + 1))
        return r2l_set_error(R2L_EINVAL, "bad argument to nerf_embed (n=%d dim=%d multires=%d ldi=%lld ldo=%lld)", n, dim, multires, ldi, ldo);
    int rc = r2l_require_gfx950(nullptr);
    if (rc) return rc;
    if (n == 0) return R2L_OK;
    const long long total = (long long)n * dim;
    hipLaunchKernelGGL(nerf_embed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x_dev, ldi, (long long)n,
                       dim, multires, out_dev, ldo);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "nerf_embed launch: %s", hipGetErrorString(e));
    return R2L_OK;
}

int r2l_linear_create(r2l_linear** out, const float* w_host, const float* b_host, int out_dim, int in_dim) {
    if (!out || !w_host || out_dim <= 0 || in_dim <= 0 || out_dim > (1 << 16) || in_dim > (1 << 16))
        return r2l_set_error(R2L_EINVAL, "bad argument to r2l_linear_create (out_dim=%d in_dim=%d)", out_dim, in_dim);
    int rc = r2l_require_gfx950(nullptr);
    if (rc) return rc;
    r2l_linear* l = new r2l_linear{in_dim, out_dim, nullptr, nullptr};
    const size_t wn = (size_t)out_dim * in_dim;
    hipError_t e = hipMalloc(&l->w_dev, wn * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&l->b_dev, (size_t)out_dim * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(l->w_dev, w_host, wn * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        std::vector<float> zero;
        if (!b_host) zero.assign(out_dim, 0.0f);
        e = hipMemcpy(l->b_dev, b_host ? b_host : zero.data(), (size_t)out_dim * sizeof(float), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        if (l->w_dev) (void)hipFree(l->w_dev);
        if (l->b_dev) (void)hipFree(l->b_dev);
        delete l;
        return r2l_set_error(R2L_EHIP, "r2l_linear_create: %s", hipGetErrorString(e));
    }
    *out = l;
    return R2L_OK;
}

void r2l_linear_destroy(r2l_linear* l) {
    if (!l) return;
    (void)hipFree(l->w_dev);
    (void)hipFree(l->b_dev);
    delete l;
}

int r2l_linear_forward(const r2l_linear* l, const float* x_dev, long long ldx, int n, float* y_dev, long long ldy,
                       const float* res_dev, long long ldr, float res_scale, int act, const float* post_dev, long long ldp,
                       void* stream) {
    if (!l || !x_dev || !y_dev || n < 0 || ldx < l->in_dim || ldy < l->out_dim || (res_dev && ldr < l->out_dim) ||
        (post_dev && ldp < l->out_dim) || act < R2L_ACT_NONE || act > R2L_ACT_SIGMOID)
        return r2l_set_error(R2L_EINVAL, "bad argument to r2l_linear_forward (n=%d ldx=%lld ldy=%lld act=%d; layer %d -> %d)", n, ldx,
                             ldy, act, l ? l->in_dim : -1, l ? l->out_dim : -1);
    // the input tile is read by every column block while others write: x and y must not overlap (res / post may be y itself:
    // an element is read and written by the same thread)
    const float* x_end = x_dev + (size_t)(n > 0 ? n - 1 : 0) * ldx + l->in_dim;
    const float* y_end = y_dev + (size_t)(n > 0 ? n - 1 : 0) * ldy + l->out_dim;
    if (n > 0 && x_dev < y_end && y_dev < x_end)
        return r2l_set_error(R2L_EINVAL, "r2l_linear_forward: x and y overlap");
    if (n == 0) return R2L_OK;
    dim3 grid((unsigned)((n + GT_M - 1) / GT_M), (unsigned)((l->out_dim + GT_N - 1) / GT_N));
    hipLaunchKernelGGL(r2l_linear_kernel, grid, dim3(256), 0, (hipStream_t)stream, x_dev, ldx, n, l->in_dim, l->w_dev, l->b_dev,
                       l->out_dim, y_dev, ldy, res_dev, ldr, res_scale, act, post_dev, ldp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "r2l_linear_forward launch: %s", hipGetErrorString(e));
    return R2L_OK;
}

int r2l_sample_points(const float* rays_o_dev, const float* rays_d_dev, int n, const float* z_dev, int n_sample, int z_per_ray,
                      float* pts_out_dev, void* stream) {
    if (!rays_o_dev || !rays_d_dev || !z_dev || !pts_out_dev || n < 0 || n_sample <= 0)
        return r2l_set_error(R2L_EINVAL, "bad argument to r2l_sample_points");
    int rc = r2l_require_gfx950(nullptr);
    if (rc) return rc;
    if (n == 0) return R2L_OK;
    const long long total = (long long)n * n_sample;
    hipLaunchKernelGGL(r2l_sample_points_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rays_o_dev,
                       rays_d_dev, (long long)n, z_dev, n_sample, z_per_ray ? 1 : 0, pts_out_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "r2l_sample_points launch: %s", hipGetErrorString(e));
    return R2L_OK;
}

}  // extern "C"
