// Which instruction of a CO-RESIDENT wave makes a packed op with a cross-half op_sel go wrong (profiles/r04_gpu_sharing.txt)?
// tools/pk_localise_probe located the wrong value of the SLP-built get_rays kernel in
//     v_pk_add_f32 D, X, Y op_sel:[0,1] op_sel_hi:[1,0]          D.lo = X.lo + Y.hi      (got: X.lo + 0 in groups of 16 lanes)
// with VGPR-only operands, and only while another process's nerf_chain_kernel (256 VGPRs + 144 AGPRs per wave: 112 registers of every
// SIMD stay free, so a small kernel's waves run ON THE SAME SIMD beside it; the R2L body kernel takes all 512 and never triggered it)
// was on the card.  This probe needs no second process: a heavy kernel loops ONE instruction class on stream A with few registers,
// the light kernels run on stream B beside it and check every packed result against scalar arithmetic on the device.
//   usage: coresidency_probe [seconds per cell] [only this heavy class]
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/coresidency_probe.hip -o tools/coresidency_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x6 __attribute__((ext_vector_type(6)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

enum { H_NONE, H_MFMA16_F16, H_MFMA16_BF6, H_MFMA32_F16, H_MFMA32_BF6, H_CVT_BF6, H_DS_READ, H_LDS_DMA, H_FMA_MIX, H_ACCVGPR, H_VFMA, H_SPIN,
       H_MFMA16_BF16, H_MFMA16_FP8, H_MFMA16_I8, H_MFMA16_F32, H_MFMA16_F64, H_MFMA4_F16, H_MFMA16K16_F16, H_PKFMA_F16, H_N };
static const char* hname[H_N] = {"(nothing beside it)", "v_mfma_f32_16x16x32_f16", "v_mfma_scale_f32_16x16x128_f8f6f4 (bf6)", "v_mfma_f32_32x32x16_f16",
                                 "v_mfma_scale_f32_32x32x64_f8f6f4 (bf6)", "v_cvt_scalef32_pk32_bf6_f16", "ds_read_b128", "global_load_lds_dwordx4",
                                 "v_fma_mixlo/hi_f16", "v_accvgpr_write / read", "v_fma_f32", "s_nop spin", "v_mfma_f32_16x16x32_bf16",
                                 "v_mfma_f32_16x16x128_f8f6f4 (fp8, no scale)", "v_mfma_i32_16x16x64_i8", "v_mfma_f32_16x16x4_f32", "v_mfma_f64_16x16x4_f64",
                                 "v_mfma_f32_4x4x4_16b_f16", "v_mfma_f32_16x16x16_f16", "v_pk_fma_f16"};

__device__ __forceinline__ unsigned mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// the heavy side: 64 instructions of one class per loop iteration, every operand random bits, 4 waves per workgroup
template <int HV>
__global__ __launch_bounds__(256) void heavy(int iters, const char* __restrict__ stream, float* sink, float* dump) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned seed = mix(blockIdx.x * 256 + threadIdx.x + 1);
    i32x6 a, b;
#pragma unroll
    for (int j = 0; j < 6; ++j) { a[j] = (int)(mix(seed + j) & 0x3bff3bffu); b[j] = (int)(mix(seed + 16 + j) & 0x3bff3bffu); }     // finite f16 pairs
    f32x16 c0 = {0}, c1 = {0};
    f32x4 h0 = {0, 0, 0, 0}, h1 = h0, d0 = h0;
    typedef double f64x4 __attribute__((ext_vector_type(4)));
    f64x4 dd0 = {0, 0, 0, 0}, dd1 = dd0;
    i32x16 src;
#pragma unroll
    for (int j = 0; j < 16; ++j) src[j] = (int)(mix(seed + 32 + j) & 0x3bff3bffu);
    i32x6 cv = {0, 0, 0, 0, 0, 0};
    int sc = 127;
    float v0 = __int_as_float(mix(seed + 77) & 0x3f7fffffu), v1 = __int_as_float(mix(seed + 78) & 0x3f7fffffu), v2 = 0.f, scale = 1.0f;
    unsigned x0 = mix(seed + 79) & 0x3bff3bffu, x1 = 0, acct = 0;
    unsigned addr = wave * 16384 + lane * 16;
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<unsigned*>(smem)[i] = mix(seed + i);
    __syncthreads();
    unsigned goff = (unsigned)(((blockIdx.x * 4 + wave) * 73 * 1024) % (16u << 20)) + lane * 16;
    const unsigned m0v = __builtin_amdgcn_readfirstlane(wave * 16384);
    for (int i = 0; i < iters; ++i) {
        if (HV == H_MFMA16_F16)
            asm volatile(".rept 32\n v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_f16 %1, %3, %2, %1\n .endr\n"
                         : "+v"(h0), "+v"(h1) : "v"(__builtin_shufflevector(a, a, 0, 1, 2, 3)), "v"(__builtin_shufflevector(b, b, 0, 1, 2, 3)));
        else if (HV == H_MFMA16_BF6)
            asm volatile(".rept 32\n v_mfma_scale_f32_16x16x128_f8f6f4 %0, %2, %3, %0, %4, %4 op_sel_hi:[0,0,0] cbsz:3 blgp:3\n"
                         " v_mfma_scale_f32_16x16x128_f8f6f4 %1, %3, %2, %1, %4, %4 op_sel_hi:[0,0,0] cbsz:3 blgp:3\n .endr\n"
                         : "+v"(h0), "+v"(h1) : "v"(a), "v"(b), "v"(sc));
        else if (HV == H_MFMA32_F16)
            asm volatile(".rept 32\n v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_f16 %1, %3, %2, %1\n .endr\n"
                         : "+v"(c0), "+v"(c1) : "v"(__builtin_shufflevector(a, a, 0, 1, 2, 3)), "v"(__builtin_shufflevector(b, b, 0, 1, 2, 3)));
        else if (HV == H_MFMA32_BF6)
            asm volatile(".rept 32\n v_mfma_scale_f32_32x32x64_f8f6f4 %0, %2, %3, %0, %4, %4 op_sel_hi:[0,0,0] cbsz:3 blgp:3\n"
                         " v_mfma_scale_f32_32x32x64_f8f6f4 %1, %3, %2, %1, %4, %4 op_sel_hi:[0,0,0] cbsz:3 blgp:3\n .endr\n"
                         : "+v"(c0), "+v"(c1) : "v"(a), "v"(b), "v"(sc));
        else if (HV == H_CVT_BF6)
            asm volatile(".rept 64\n v_cvt_scalef32_pk32_bf6_f16 %0, %1, %2\n .endr\n" : "+v"(cv) : "v"(src), "v"(scale));
        else if (HV == H_DS_READ)
            asm volatile(".set cp_s, 0\n .rept 64\n ds_read_b128 %0, %1 offset:(1024 * (cp_s & 15))\n .set cp_s, cp_s + 1\n .endr\n s_waitcnt lgkmcnt(0)\n"
                         : "+v"(d0) : "v"(addr) : "memory");
        else if (HV == H_LDS_DMA)
            asm volatile("s_mov_b32 m0, %2\n s_nop 0\n .rept 16\n global_load_lds_dwordx4 %0, %1\n v_add_u32 %0, 0x400, %0\n .endr\n"
                         "v_and_b32 %0, 0xffffff, %0\n s_waitcnt vmcnt(0)\n"
                         : "+v"(goff) : "s"(stream), "s"(m0v) : "memory");
        else if (HV == H_FMA_MIX)
            asm volatile(".rept 32\n v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n v_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n .endr\n"
                         : "+v"(x1) : "v"(x0), "v"(scale), "v"(v0), "v"(v1));
        else if (HV == H_ACCVGPR)
            asm volatile(".rept 32\n v_accvgpr_write_b32 a0, %1\n s_nop 0\n v_accvgpr_read_b32 %0, a0\n .endr\n" : "+v"(acct) : "v"(v0) : "a0");
        else if (HV == H_VFMA)
            asm volatile(".rept 64\n v_fma_f32 %0, %1, %2, %0\n .endr\n" : "+v"(v2) : "v"(v0), "v"(v1));
        else if (HV == H_MFMA16_BF16)
            asm volatile(".rept 32\n v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_bf16 %1, %3, %2, %1\n .endr\n"
                         : "+v"(h0), "+v"(h1) : "v"(__builtin_shufflevector(a, a, 0, 1, 2, 3)), "v"(__builtin_shufflevector(b, b, 0, 1, 2, 3)));
        else if (HV == H_MFMA16_FP8)
            asm volatile(".rept 32\n v_mfma_f32_16x16x128_f8f6f4 %0, %2, %3, %0\n v_mfma_f32_16x16x128_f8f6f4 %1, %3, %2, %1\n .endr\n"
                         : "+v"(h0), "+v"(h1) : "v"(__builtin_shufflevector(src, src, 0, 1, 2, 3, 4, 5, 6, 7)), "v"(__builtin_shufflevector(src, src, 8, 9, 10, 11, 12, 13, 14, 15)));
        else if (HV == H_MFMA16_I8)
            asm volatile(".rept 32\n v_mfma_i32_16x16x64_i8 %0, %2, %3, %0\n v_mfma_i32_16x16x64_i8 %1, %3, %2, %1\n .endr\n"
                         : "+v"(h0), "+v"(h1) : "v"(__builtin_shufflevector(a, a, 0, 1, 2, 3)), "v"(__builtin_shufflevector(b, b, 0, 1, 2, 3)));
        else if (HV == H_MFMA16_F32)
            asm volatile(".rept 32\n v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n v_mfma_f32_16x16x4_f32 %1, %3, %2, %1\n .endr\n"
                         : "+v"(h0), "+v"(h1) : "v"(v0), "v"(v1));
        else if (HV == H_MFMA16_F64)
            asm volatile(".rept 32\n v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n v_mfma_f64_16x16x4_f64 %1, %3, %2, %1\n .endr\n"
                         : "+v"(dd0), "+v"(dd1) : "v"(1.0 + (double)v0), "v"(0.5 + (double)v1));
        else if (HV == H_MFMA4_F16)
            asm volatile(".rept 32\n v_mfma_f32_4x4x4_16b_f16 %0, %2, %3, %0\n v_mfma_f32_4x4x4_16b_f16 %1, %3, %2, %1\n .endr\n"
                         : "+v"(h0), "+v"(h1) : "v"(__builtin_shufflevector(a, a, 0, 1)), "v"(__builtin_shufflevector(b, b, 0, 1)));
        else if (HV == H_MFMA16K16_F16)
            asm volatile(".rept 32\n v_mfma_f32_16x16x16_f16 %0, %2, %3, %0\n v_mfma_f32_16x16x16_f16 %1, %3, %2, %1\n .endr\n"
                         : "+v"(h0), "+v"(h1) : "v"(__builtin_shufflevector(a, a, 0, 1)), "v"(__builtin_shufflevector(b, b, 0, 1)));
        else if (HV == H_PKFMA_F16)
            asm volatile(".rept 64\n v_pk_fma_f16 %0, %1, %2, %0\n .endr\n" : "+v"(x1) : "v"(x0), "v"(a[0]));
        else
            asm volatile(".rept 64\n s_nop 7\n .endr\n");
    }
    float s = c0[0] + c1[1] + h0[2] + h1[3] + v2 + d0[0] + (float)(dd0[0] + dd1[1]) + __int_as_float(x1 ^ acct ^ cv[0] ^ cv[5] ^ goff);
    if (s == 12345.678f) sink[0] = s;
    if (dump) {      // the heavy side's own results, for the bitwise comparison alone / beside the light kernels
        float* d = dump + (size_t)(blockIdx.x * 256 + threadIdx.x) * 12;
#pragma unroll
        for (int j = 0; j < 4; ++j) { d[j] = h0[j]; d[4 + j] = h1[j]; d[8 + j] = c0[j] + c1[j + 4]; }
    }
}

enum { L_PKADD_X1, L_PKADD, L_PKADD_X0, L_PKMUL_X1, L_PKFMA_X1, L_PKADD_F16_X1, L_SCALAR, L_PKADD_HH, L_PKADD_LOX, L_PKADD_HIX, L_PKFMA_X2, L_SAMEWAVE, L_PKADD_X1_NOP, L_N };
static const char* lname[L_N] = {"v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]  (lo = X.lo + Y.hi, hi = X.hi + Y.lo)",
                                 "v_pk_add_f32 (no cross-half select)",
                                 "v_pk_add_f32 op_sel:[1,0] op_sel_hi:[0,1]  (lo = X.hi + Y.lo, hi = X.lo + Y.hi)",
                                 "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]",
                                 "v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1]",
                                 "v_pk_add_f16 op_sel:[0,1] op_sel_hi:[1,0]",
                                 "v_add_f32 x 2 (scalar control)",
                                 "v_pk_add_f32 op_sel:[1,1] op_sel_hi:[1,1]  (lo = hi = X.hi + Y.hi)",
                                 "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,1]  (lo = X.lo + Y.hi, hi = X.hi + Y.hi)",
                                 "v_pk_add_f32 op_sel:[0,0] op_sel_hi:[1,0]  (lo = X.lo + Y.lo, hi = X.hi + Y.lo)",
                                 "v_pk_fma_f32 op_sel:[0,0,1] op_sel_hi:[1,1,0]  (src2 crosses)",
                                 "v_mfma_f32_16x16x32_f16 in the SAME wave right in front of the crossing v_pk_add_f32",
                                 "s_nop 7 in front of and behind the crossing v_pk_add_f32"};

// the light side: 256 rounds of one packed op per thread, each checked against scalar arithmetic; cnt[0] ops, [1] wrong lo, [2] wrong hi,
// [3] bit mask of the 16-lane groups (lane >> 4) seen wrong
template <int LV>
__global__ __launch_bounds__(256) void light(int n, unsigned long long* cnt, int first_launch, f32x2* seen) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long bad_lo = 0, bad_hi = 0;
    unsigned grp = 0;
    for (int r = 0; r < 256; ++r) {
        // small integers / 8: every sum and product below is exact in fp32 (and the f16 ones in f16)
        const float xa = (float)((i * 7 + r * 3) & 255) * 0.125f + 1.0f, xb = (float)((i * 5 + r) & 127) * 0.25f - 9.0f;
        const float ya = (float)((i * 3 + r * 11) & 255) * 0.5f - 17.0f, yb = (float)((i + r * 13) & 63) * 0.125f + 0.5f;
        const f32x2 X = {xa, xb}, Y = {ya, yb}, Z = {3.0f, -5.0f};
        f32x2 D, E;
        if (LV == L_PKADD_X1) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(D) : "v"(X), "v"(Y));
            E[0] = __fadd_rn(xa, yb); E[1] = __fadd_rn(xb, ya);
        } else if (LV == L_PKADD) {
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(D) : "v"(X), "v"(Y));
            E[0] = __fadd_rn(xa, ya); E[1] = __fadd_rn(xb, yb);
        } else if (LV == L_PKADD_X0) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(D) : "v"(X), "v"(Y));
            E[0] = __fadd_rn(xb, ya); E[1] = __fadd_rn(xa, yb);
        } else if (LV == L_PKMUL_X1) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(D) : "v"(X), "v"(Y));
            E[0] = __fmul_rn(xa, yb); E[1] = __fmul_rn(xb, ya);
        } else if (LV == L_PKFMA_X1) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(D) : "v"(X), "v"(Y), "v"(Z));
            E[0] = __fmaf_rn(xa, yb, 3.0f); E[1] = __fmaf_rn(xb, ya, -5.0f);
        } else if (LV == L_PKADD_F16_X1) {
            const f16x2 hx = {(_Float16)(float)((i + r) & 31), (_Float16)(float)(((i >> 3) + r) & 15)};
            const f16x2 hy = {(_Float16)(float)((i * 3 + r) & 63), (_Float16)(-(float)((i + r * 5) & 7))};
            f16x2 hd;
            asm volatile("v_pk_add_f16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(hd) : "v"(hx), "v"(hy));
            D[0] = (float)hd[0]; D[1] = (float)hd[1];
            E[0] = (float)hx[0] + (float)hy[1]; E[1] = (float)hx[1] + (float)hy[0];
        } else if (LV == L_SCALAR) {
            asm volatile("v_add_f32 %0, %2, %3\n v_add_f32 %1, %4, %5" : "=&v"(D[0]), "=&v"(D[1]) : "v"(xa), "v"(yb), "v"(xb), "v"(ya));
            E[0] = __fadd_rn(xa, yb); E[1] = __fadd_rn(xb, ya);
        } else if (LV == L_PKADD_HH) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1]" : "=v"(D) : "v"(X), "v"(Y));
            E[0] = __fadd_rn(xb, yb); E[1] = E[0];
        } else if (LV == L_PKADD_LOX) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(D) : "v"(X), "v"(Y));
            E[0] = __fadd_rn(xa, yb); E[1] = __fadd_rn(xb, yb);
        } else if (LV == L_PKADD_HIX) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(D) : "v"(X), "v"(Y));
            E[0] = __fadd_rn(xa, ya); E[1] = __fadd_rn(xb, ya);
        } else if (LV == L_PKFMA_X2) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(D) : "v"(X), "v"(Y), "v"(Z));
            E[0] = __fmaf_rn(xa, ya, -5.0f); E[1] = __fmaf_rn(xb, yb, 3.0f);
        } else if (LV == L_SAMEWAVE) {
            f32x4 acc = {0, 0, 0, 0};
            const f32x4 fa = {xa, xb, ya, yb};
            asm volatile("v_mfma_f32_16x16x32_f16 %1, %4, %4, %1\n v_pk_add_f32 %0, %2, %3 op_sel:[0,1] op_sel_hi:[1,0]\n s_nop 7\n s_nop 7"
                         : "=&v"(D), "+v"(acc) : "v"(X), "v"(Y), "v"(fa));
            E[0] = __fadd_rn(xa, yb); E[1] = __fadd_rn(xb, ya);
            if (acc[0] == 12345.678f) ++bad_hi;
        } else {
            asm volatile("s_nop 7\n v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]\n s_nop 7" : "=v"(D) : "v"(X), "v"(Y));
            E[0] = __fadd_rn(xa, yb); E[1] = __fadd_rn(xb, ya);
        }
        if (__float_as_uint(D[0]) != __float_as_uint(E[0])) { ++bad_lo; grp |= 1u << ((threadIdx.x & 63) >> 4); }
        if (__float_as_uint(D[1]) != __float_as_uint(E[1])) { ++bad_hi; grp |= 1u << (4 + ((threadIdx.x & 63) >> 4)); }
    }
    if (bad_lo) atomicAdd(&cnt[1], bad_lo);
    if (bad_hi) atomicAdd(&cnt[2], bad_hi);
    if (grp) atomicOr(&cnt[3], (unsigned long long)grp);
    if (threadIdx.x == 0) atomicAdd(&cnt[0], 256ull * blockDim.x);
}

static float* g_dump = nullptr;
template <int HV> static void launch_heavy(hipStream_t s, int iters, const char* stream, float* sink) {
    hipLaunchKernelGGL(heavy<HV>, dim3(512), dim3(256), 65536, s, iters, stream, sink, g_dump);
}
static void heavy_any(int hv, hipStream_t s, int iters, const char* stream, float* sink) {
    switch (hv) {
        case H_MFMA16_F16: launch_heavy<H_MFMA16_F16>(s, iters, stream, sink); break;
        case H_MFMA16_BF6: launch_heavy<H_MFMA16_BF6>(s, iters, stream, sink); break;
        case H_MFMA32_F16: launch_heavy<H_MFMA32_F16>(s, iters, stream, sink); break;
        case H_MFMA32_BF6: launch_heavy<H_MFMA32_BF6>(s, iters, stream, sink); break;
        case H_CVT_BF6: launch_heavy<H_CVT_BF6>(s, iters, stream, sink); break;
        case H_DS_READ: launch_heavy<H_DS_READ>(s, iters, stream, sink); break;
        case H_LDS_DMA: launch_heavy<H_LDS_DMA>(s, iters, stream, sink); break;
        case H_FMA_MIX: launch_heavy<H_FMA_MIX>(s, iters, stream, sink); break;
        case H_ACCVGPR: launch_heavy<H_ACCVGPR>(s, iters, stream, sink); break;
        case H_VFMA: launch_heavy<H_VFMA>(s, iters, stream, sink); break;
        case H_SPIN: launch_heavy<H_SPIN>(s, iters, stream, sink); break;
        case H_MFMA16_BF16: launch_heavy<H_MFMA16_BF16>(s, iters, stream, sink); break;
        case H_MFMA16_FP8: launch_heavy<H_MFMA16_FP8>(s, iters, stream, sink); break;
        case H_MFMA16_I8: launch_heavy<H_MFMA16_I8>(s, iters, stream, sink); break;
        case H_MFMA16_F32: launch_heavy<H_MFMA16_F32>(s, iters, stream, sink); break;
        case H_MFMA16_F64: launch_heavy<H_MFMA16_F64>(s, iters, stream, sink); break;
        case H_MFMA4_F16: launch_heavy<H_MFMA4_F16>(s, iters, stream, sink); break;
        case H_MFMA16K16_F16: launch_heavy<H_MFMA16K16_F16>(s, iters, stream, sink); break;
        case H_PKFMA_F16: launch_heavy<H_PKFMA_F16>(s, iters, stream, sink); break;
        default: break;
    }
}
template <int LV> static void launch_light(hipStream_t s, int n, unsigned long long* cnt, int first, f32x2* seen) {
    hipLaunchKernelGGL(light<LV>, dim3((n + 255) / 256), dim3(256), 0, s, n, cnt, first, seen);
}
static void light_any(int lv, hipStream_t s, int n, unsigned long long* cnt, int first, f32x2* seen) {
    switch (lv) {
        case L_PKADD_X1: launch_light<L_PKADD_X1>(s, n, cnt, first, seen); break;
        case L_PKADD: launch_light<L_PKADD>(s, n, cnt, first, seen); break;
        case L_PKADD_X0: launch_light<L_PKADD_X0>(s, n, cnt, first, seen); break;
        case L_PKMUL_X1: launch_light<L_PKMUL_X1>(s, n, cnt, first, seen); break;
        case L_PKFMA_X1: launch_light<L_PKFMA_X1>(s, n, cnt, first, seen); break;
        case L_PKADD_F16_X1: launch_light<L_PKADD_F16_X1>(s, n, cnt, first, seen); break;
        case L_SCALAR: launch_light<L_SCALAR>(s, n, cnt, first, seen); break;
        case L_PKADD_HH: launch_light<L_PKADD_HH>(s, n, cnt, first, seen); break;
        case L_PKADD_LOX: launch_light<L_PKADD_LOX>(s, n, cnt, first, seen); break;
        case L_PKADD_HIX: launch_light<L_PKADD_HIX>(s, n, cnt, first, seen); break;
        case L_PKFMA_X2: launch_light<L_PKFMA_X2>(s, n, cnt, first, seen); break;
        case L_SAMEWAVE: launch_light<L_SAMEWAVE>(s, n, cnt, first, seen); break;
        default: launch_light<L_PKADD_X1_NOP>(s, n, cnt, first, seen); break;
    }
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 1.0;
    const int only = argc > 2 ? atoi(argv[2]) : -1;
    const int n = 160000;
    hipStream_t sa, sb;
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    char* stream;
    float* sink;
    unsigned long long* cnt;
    f32x2* seen;
    CHECK(hipMalloc(&stream, 20u << 20));
    CHECK(hipMemset(stream, 0x3c, 20u << 20));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMalloc(&cnt, L_N * 4 * sizeof(unsigned long long)));
    CHECK(hipMalloc(&seen, 64 * sizeof(f32x2)));
    std::vector<unsigned long long> h(L_N * 4);
    for (int hv = 0; hv < H_N; ++hv) {
        if (only >= 0 && hv != only) continue;
        CHECK(hipMemset(cnt, 0, L_N * 4 * sizeof(unsigned long long)));
        // heavy launches of about 10 ms each, kept two deep on stream A while stream B cycles through the light kernels
        int iters = 2000;
        if (hv != H_NONE) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            heavy_any(hv, sa, 200, stream, sink);
            hipEventRecord(e0, sa);
            heavy_any(hv, sa, 200, stream, sink);
            hipEventRecord(e1, sa);
            CHECK(hipStreamSynchronize(sa));
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            iters = (int)(200 * 10.0 / (ms > 1e-3 ? ms : 1e-3));
            if (iters < 1) iters = 1;
        }
        const auto t0 = std::chrono::steady_clock::now();
        hipEvent_t hev[2];
        hipEventCreate(&hev[0]); hipEventCreate(&hev[1]);
        int hl = 0;
        long long rounds = 0;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
            if (hv != H_NONE) {
                if (hl >= 2) hipEventSynchronize(hev[hl & 1]);
                heavy_any(hv, sa, iters, stream, sink);
                hipEventRecord(hev[hl & 1], sa);
                ++hl;
            }
            for (int rep = 0; rep < 4; ++rep)
                for (int lv = 0; lv < L_N; ++lv) light_any(lv, sb, n, cnt + 4 * lv, 0, seen);
            CHECK(hipStreamSynchronize(sb));
            ++rounds;
        }
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), cnt, L_N * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        printf("beside %-44s (%lld rounds, heavy launches %d x %d iterations)\n", hname[hv], rounds, hl, iters);
        for (int lv = 0; lv < L_N; ++lv)
            printf("    %-88s %12llu ops   wrong lo %10llu   wrong hi %10llu   16-lane groups seen wrong: lo %x hi %x\n", lname[lv], h[4 * lv],
                   h[4 * lv + 1], h[4 * lv + 2], (unsigned)(h[4 * lv + 3] & 15), (unsigned)((h[4 * lv + 3] >> 4) & 15));
        fflush(stdout);
    }
    // the heavy side's own arithmetic: MFMA results of a launch alone against launches with the crossing packed ops beside them
    if (only < 0) {
        const size_t dn = (size_t)512 * 256 * 12;
        float* dump;
        CHECK(hipMalloc(&dump, dn * 4));
        std::vector<float> ref(dn), got(dn);
        const int hvs[4] = {H_MFMA16_F16, H_MFMA32_F16, H_MFMA16_BF6, H_MFMA32_BF6};
        for (int q = 0; q < 4; ++q) {
            g_dump = dump;
            CHECK(hipMemsetAsync(dump, 0, dn * 4, sa));
            heavy_any(hvs[q], sa, 3000, stream, sink);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(ref.data(), dump, dn * 4, hipMemcpyDeviceToHost));
            const char* wn[4] = {"alone again", "beside scalar v_add_f32 kernels", "beside v_pk_add_f32 without a cross-half select", "beside the crossing v_pk_add_f32 / v_pk_fma_f32"};
            for (int w = 0; w < 4; ++w) {
                long long diff = 0, launches = 0;
                int shown = 0;
                for (int rep = 0; rep < 20; ++rep) {
                    CHECK(hipMemsetAsync(dump, 0, dn * 4, sa));      // on the heavy stream: the streams are non-blocking
                    heavy_any(hvs[q], sa, 3000, stream, sink);
                    for (int k = 0; k < 6 && w > 0; ++k) {
                        light_any(w == 1 ? L_SCALAR : w == 2 ? L_PKADD : L_PKADD_X1, sb, n, cnt, 0, seen);
                        light_any(w == 1 ? L_SCALAR : w == 2 ? L_PKADD : L_PKFMA_X1, sb, n, cnt + 4, 0, seen);
                    }
                    CHECK(hipDeviceSynchronize());
                    CHECK(hipMemcpy(got.data(), dump, dn * 4, hipMemcpyDeviceToHost));
                    long long d1 = 0;
                    for (size_t i = 0; i < dn; ++i)
                        if (memcmp(&got[i], &ref[i], 4)) {
                            ++d1;
                            if (shown < 4) {
                                ++shown;
                                printf("        launch %d thread %zu (workgroup %zu, wave %zu, lane %zu) value %zu: %.9g against %.9g alone\n", rep, i / 12,
                                       i / 12 / 256, (i / 12 % 256) / 64, i / 12 % 64, i % 12, got[i], ref[i]);
                            }
                        }
                    diff += d1;
                    launches += d1 != 0;
                }
                printf("%-44s its own results, 20 launches %-50s %8lld of %zu values differ from the first launch (in %lld launches)\n",
                       hname[hvs[q]], wn[w], diff, 20 * dn, launches);
                fflush(stdout);
            }
        }
        g_dump = nullptr;
    }
    return 0;
}
