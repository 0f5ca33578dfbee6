// Does a straight-line loop larger than the 64 KiB instruction cache still run at MFMA speed on gfx950?
// One wave per SIMD on every CU executes `iters` passes over a block of REPT x (1 MFMA + 2 VALU) = REPT x 16 bytes at
// 1 byte of code per MFMA-pipe cycle (the density of the generated teacher chain).  Prints cycles per 16-byte group:
// 16 = the MFMA pipe is the limit, more = instruction fetch is.
//   hipcc --offload-arch=gfx950 -O3 tools/icache_test.hip -o tools/icache_test && tools/icache_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int REPT>
__global__ __launch_bounds__(256, 1) void k(int iters, long long* out, float* sink) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    f16x8 x = {1, 1, 1, 1, 1, 1, 1, 1};
    float v0 = threadIdx.x, v1 = 1.f;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        asm volatile(
            ".rept %c[n]\n"
            "v_mfma_f32_16x16x32_f16 %[a0], %[x], %[x], %[a0]\n v_add_f32 %[v0], %[v0], %[v1]\n v_add_f32 %[v1], %[v1], %[v0]\n"
            "v_mfma_f32_16x16x32_f16 %[a1], %[x], %[x], %[a1]\n v_add_f32 %[v0], %[v0], %[v1]\n v_add_f32 %[v1], %[v1], %[v0]\n"
            "v_mfma_f32_16x16x32_f16 %[a2], %[x], %[x], %[a2]\n v_add_f32 %[v0], %[v0], %[v1]\n v_add_f32 %[v1], %[v1], %[v0]\n"
            "v_mfma_f32_16x16x32_f16 %[a3], %[x], %[x], %[a3]\n v_add_f32 %[v0], %[v0], %[v1]\n v_add_f32 %[v1], %[v1], %[v0]\n"
            ".endr\n"
            : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [v0] "+v"(v0), [v1] "+v"(v1)
            : [x] "v"(x), [n] "i"(REPT / 4));
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (v0 == 12345.f) sink[0] = a0[0] + a1[0] + a2[0] + a3[0] + v1;
}

template <int REPT>
void run(long long* d_out, float* d_sink, int nblk) {
    int iters = (1 << 22) / REPT;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<REPT><<<nblk, 256>>>(2, d_out, d_sink);
    hipEventRecord(e0);
    k<REPT><<<nblk, 256>>>(iters, d_out, d_sink);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(nblk);
    hipMemcpy(h.data(), d_out, nblk * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += v; mean /= nblk;
    printf("code %4d KiB  iters %6d  %.3f ms  ns per group %.2f  (shader clock ticks per group %.2f)\n", REPT * 16 / 1024, iters, ms,
           ms * 1e6 / ((double)iters * REPT), mean / ((double)iters * REPT));
}

int main() {
    long long* d_out; float* d_sink;
    hipMalloc(&d_out, 1024 * 8); hipMalloc(&d_sink, 64);
    for (int nblk : {256, 512}) {
        printf("workgroups %d (4 waves each)\n", nblk);
        run<1024>(d_out, d_sink, nblk);
        run<2048>(d_out, d_sink, nblk);
        run<3072>(d_out, d_sink, nblk);
        run<4096>(d_out, d_sink, nblk);
        run<5120>(d_out, d_sink, nblk);
        run<6144>(d_out, d_sink, nblk);
        run<7168>(d_out, d_sink, nblk);
    }
    return 0;
}
