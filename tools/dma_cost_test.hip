// Probe: what one 1 KiB/wave weight-staging step costs the issuing wave inside an MFMA stream
// (one wave per SIMD, 16x16x32 f16 MFMAs, 4 independent accumulators):
//   mode 0: MFMAs only          mode 1: + global_load_lds_dwordx4 (LDS-DMA) every 4 MFMAs
//   mode 2: + global_load_dwordx4 -> registers -> ds_write_b128 (one group later) every 4 MFMAs
//   mode 3: + global_load_dwordx4 only (data dropped)      mode 4: + ds_write_b128 only
//   mode 5 / 6: + one / two ds_read_b128
// hipcc --offload-arch=gfx950 -O3 tools/dma_cost_test.hip -o tools/dma_cost_test
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) char smem[];
#define AS1(p) ((const __attribute__((address_space(1))) void*)(p))
#define AS3(p) ((__attribute__((address_space(3))) void*)(p))

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const char* __restrict__ w, float* out, int iters, long long* cyc) {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x + i); b[i] = (_Float16)(i); }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const char* src = w + (size_t)blockIdx.x * 65536 + wave * 16384 + lane * 16;
    i32x4 stage[8], stage2[8];
    for (int i = 0; i < 8; ++i) stage2[i] = i32x4{0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) stage[i] = i32x4{0, 0, 0, 0};
    long long t0 = clock64();
    for (int it = 0; it < iters; it += 2) {
        const int slot = (it & 15) * 1024;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[3], 0, 0, 0);
            const uint32_t lds = wave * 32768 + ((slot + g * 4096) & 32767) + 0;
            if (MODE == 1) __builtin_amdgcn_global_load_lds(AS1(src + ((slot + g * 4096) & 16383)), AS3(smem + lds), 16, 0, 0);
            // the register ring is 8 deep: a load is consumed 8 groups (~800 cycles) after it was issued
            if (MODE == 2 || MODE == 4) *reinterpret_cast<i32x4*>(smem + lds + lane * 16) = stage[g];
            if (MODE == 3) asm volatile("" ::"v"(stage[g]));
            if (MODE == 2 || MODE == 3) stage[g] = *reinterpret_cast<const i32x4*>(src + ((slot + g * 4096) & 16383));
            if (MODE == 5 || MODE == 6) {  // ds_read_b128 consumed 8 groups later (mode 6: two per group)
                asm volatile("" ::"v"(stage[g]));
                stage[g] = *reinterpret_cast<const i32x4*>(smem + lds + lane * 16);
                if (MODE == 6) {
                    asm volatile("" ::"v"(stage2[g]));
                    stage2[g] = *reinterpret_cast<const i32x4*>(smem + lds + 1024 + lane * 16);
                }
            }
        }
        if (MODE == 1 && (it & 2) == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    long long t1 = clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    s += *reinterpret_cast<float*>(smem + threadIdx.x * 4) + stage[0][0] + stage[3][1] + stage[7][2] + stage2[1][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    char* w; float* o; long long* c;
    hipMalloc(&w, 256 * 65536); hipMemset(w, 0, 256 * 65536); hipMalloc(&o, 256 * 256 * 4); hipMalloc(&c, 8);
    const int iters = 20000;
    for (int mode = 0; mode < 7; ++mode) {
        float ms = 0; long long cyc = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
#define L(M) hipFuncSetAttribute((const void*)k<M>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); hipLaunchKernelGGL(k<M>, dim3(256), dim3(256), 131072, 0, w, o, iters, c)
            if (mode == 0) { L(0); } else if (mode == 1) { L(1); } else if (mode == 2) { L(2); } else if (mode == 3) { L(3); } else if (mode == 4) { L(4); } else if (mode == 5) { L(5); } else { L(6); }
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(&cyc, c, 8, hipMemcpyDeviceToHost);
        }
        printf("mode %d: %.3f ms, %.1f s_memtime ticks per group of 4 MFMAs (%.1f ns)\n", mode, ms, (double)cyc / iters / 4, ms * 1e6 / iters / 4);
    }
    return 0;
}
