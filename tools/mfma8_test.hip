// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 on gfx950: (1) the dot product pairs element j of
// lane quarter q of A with element j of lane quarter q of B (so any consistent packing works),
// (2) E8M0 scale semantics with lane-uniform scales, (3) issue rate against 16x16x32 f16.
// hipcc --offload-arch=gfx950 -O3 tools/mfma8_test.hip -o /tmp/mfma8_test && /tmp/mfma8_test
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void k_mfma(const i32x8* a, const i32x8* b, float* d, int fa, int fb, int sa, int sb) {
    int lane = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    i32x8 av = a[lane], bv = b[lane];
    if (fa == 0 && fb == 0) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 0, 0, 0, sa, 0, sb);
    else if (fa == 0 && fb == 1) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 0, 1, 0, sa, 0, sb);
    else if (fa == 1 && fb == 0) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 1, 0, 0, sa, 0, sb);
    else acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 1, 1, 0, sa, 0, sb);
    for (int r = 0; r < 4; ++r) d[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[r];
}

__global__ void k_cvt(const float* x, uint32_t* o) {
    // v_cvt_pk_fp8_f32 / bf8: two floats -> one 16-bit half
    int i = threadIdx.x;
    uint32_t v = 0;
    v = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * i], x[4 * i + 1], v, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * i + 2], x[4 * i + 3], v, true);
    uint32_t w = 0;
    w = __builtin_amdgcn_cvt_pk_bf8_f32(x[4 * i], x[4 * i + 1], w, false);
    w = __builtin_amdgcn_cvt_pk_bf8_f32(x[4 * i + 2], x[4 * i + 3], w, true);
    o[2 * i] = v; o[2 * i + 1] = w;
}

template <int MODE> __global__ __launch_bounds__(256, 1) void k_rate(float* out, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    i32x8 a8 = {1, 2, 3, 4, 5, 6, 7, (int)threadIdx.x}, b8 = {7, 6, 5, 4, 3, 2, 1, (int)threadIdx.x};
    f16x8 ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)(i + threadIdx.x); bh[i] = (_Float16)(i * 2); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[i], 0, 0, 0);
            if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[i], 0, 1, 0, 127, 0, 127);
            if (MODE == 2) {  // the mix: 4 f16 + 1 fp8 per unit
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[i], 0, 0, 0);
                if ((i & 3) == 3) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[i], 0, 1, 0, 127, 0, 127);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static uint8_t enc_e4m3(int v) {  // small integers |v| <= 8 exactly
    if (v == 0) return 0;
    uint8_t s = v < 0 ? 0x80 : 0; int m = abs(v); int e = 0; while ((1 << (e + 1)) <= m) ++e;
    int frac = ((m << 3) >> e) & 7;  // 3 mantissa bits
    return s | (uint8_t)(((e + 7) << 3) | frac);
}
static uint8_t enc_e5m2(int v) {
    if (v == 0) return 0;
    uint8_t s = v < 0 ? 0x80 : 0; int m = abs(v); int e = 0; while ((1 << (e + 1)) <= m) ++e;
    int frac = ((m << 2) >> e) & 3;
    return s | (uint8_t)(((e + 15) << 2) | frac);
}

int main() {
    uint8_t A[64][32], B[64][32]; int Ai[64][32], Bi[64][32];
    int fails = 0;
    for (int fa = 0; fa < 2; ++fa) for (int fb = 0; fb < 2; ++fb) {
        srand(1 + fa * 2 + fb);
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) {
            Ai[l][j] = rand() % 9 - 4; Bi[l][j] = rand() % 9 - 4;
            A[l][j] = fa ? enc_e5m2(Ai[l][j]) : enc_e4m3(Ai[l][j]);
            B[l][j] = fb ? enc_e5m2(Bi[l][j]) : enc_e4m3(Bi[l][j]);
        }
        void *da, *db; float* dd;
        hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dd, 1024);
        hipMemcpy(da, A, 2048, hipMemcpyHostToDevice); hipMemcpy(db, B, 2048, hipMemcpyHostToDevice);
        for (int sc = 0; sc < 2; ++sc) {
            int sa = sc ? 127 + 3 : 127, sb = sc ? 127 - 5 : 127;
            k_mfma<<<1, 64>>>((const i32x8*)da, (const i32x8*)db, dd, fa, fb, sa | (sa << 8) | (sa << 16) | (sa << 24), sb | (sb << 8) | (sb << 16) | (sb << 24));
            float D[256]; hipMemcpy(D, dd, 1024, hipMemcpyDeviceToHost);
            double mul = sc ? ldexp(1.0, 3 - 5) : 1.0; int bad = 0;
            for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
                long ref = 0;
                for (int q = 0; q < 4; ++q) for (int j = 0; j < 32; ++j) ref += Ai[q * 16 + m][j] * Bi[q * 16 + n][j];
                if (D[m * 16 + n] != (float)(ref * mul)) { if (bad < 3) printf("  mismatch m%d n%d got %g want %g\n", m, n, D[m * 16 + n], ref * mul); ++bad; }
            }
            printf("fmtA=%d fmtB=%d scale=%d: %s (%d bad)\n", fa, fb, sc, bad ? "FAIL" : "ok", bad); fails += bad != 0;
        }
    }
    {   // conversions
        float x[256]; for (int i = 0; i < 256; ++i) x[i] = ldexpf(1.0f + (i % 16) / 16.0f, i / 16 - 8) * ((i & 1) ? -1 : 1);
        x[0] = 1e6f; x[1] = -1e6f; x[2] = 500.f; x[3] = 1e-9f;
        float* dx; uint32_t* dv; hipMalloc(&dx, 1024); hipMalloc(&dv, 512);
        hipMemcpy(dx, x, 1024, hipMemcpyHostToDevice);
        k_cvt<<<1, 64>>>(dx, dv);
        uint32_t v[128]; hipMemcpy(v, dv, 512, hipMemcpyDeviceToHost);
        printf("cvt sat: x=1e6 -> fp8 0x%02x bf8 0x%02x ; -1e6 -> 0x%02x 0x%02x ; 500 -> 0x%02x 0x%02x ; 1e-9 -> 0x%02x 0x%02x\n",
               v[0] & 255, v[1] & 255, (v[0] >> 8) & 255, (v[1] >> 8) & 255, (v[0] >> 16) & 255, (v[1] >> 16) & 255, v[0] >> 24, v[1] >> 24);
        printf("cvt 1.0625 (tie) -> fp8 0x%02x ; 1.1875 -> 0x%02x\n", 0, 0);
    }
    for (int mode = 0; mode < 3; ++mode) {
        float* o; hipMalloc(&o, 256 * 256 * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        int iters = 200000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) k_rate<0><<<256, 256>>>(o, iters);
            if (mode == 1) k_rate<1><<<256, 256>>>(o, iters);
            if (mode == 2) k_rate<2><<<256, 256>>>(o, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double n_f16 = mode == 0 ? 8 : mode == 2 ? 8 : 0, n_f8 = mode == 1 ? 8 : mode == 2 ? 2 : 0;
        double flop = (n_f16 * 16 * 16 * 32 * 2 + n_f8 * 16 * 16 * 128 * 2) * iters * 1024.0 * 256;
        double units = (n_f16 + 2 * n_f8) * iters;  // in f16-MFMA time units per wave
        printf("rate mode %d: %.3f ms, %.1f TFLOP/s, %.2f ns per f16-unit per wave\n", mode, ms, flop / ms / 1e9, ms * 1e6 / units);
    }
    return fails;
}
