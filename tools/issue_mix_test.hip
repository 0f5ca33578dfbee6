// How much issue time does an MFMA instruction cost next to the fillers of the generated kernels?  One wave per SIMD
// runs groups of {MFMA work of 32 pipe cycles + NV VALU + ND ds_read_b128 + NW counted waits}; the MFMA work is either
// two v_mfma_f32_16x16x32_f16 (shape 0) or one v_mfma_f32_32x32x16_f16 (shape 1).  Prints cycles per group: 32 = the
// matrix pipe is the limit.
//   hipcc --offload-arch=gfx950 -O3 tools/issue_mix_test.hip -o tools/issue_mix_test && tools/issue_mix_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
extern __shared__ char smem[];

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

template <int SHAPE, int NV, int ND, int NW>
__global__ __launch_bounds__(256, 1) void k(int iters, float* sink) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0;
    f32x16 b0 = {0};
    f16x8 x = {1, 1, 1, 1, 1, 1, 1, 1};
    typedef int i32x6 __attribute__((ext_vector_type(6)));
    i32x6 y = {0, 0, 0, 0, 0, 0};
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f;
    f32x4 d0, d1, d2, d3;
    unsigned addr = (threadIdx.x & 63) * 16;
    for (int i = 0; i < iters; ++i) {
        asm volatile(
            ".rept 64\n"
            ".if %c[shape] == 0\n"
            "v_mfma_f32_16x16x32_f16 %[a0], %[x], %[x], %[a0]\n"
            ".if %c[nv] > 0\n v_add_f32 %[v0], %[v0], %[v1]\n .endif\n"
            ".if %c[nd] > 0\n ds_read_b128 %[d0], %[addr]\n .endif\n"
            ".if %c[nv] > 1\n v_add_f32 %[v1], %[v1], %[v2]\n .endif\n"
            "v_mfma_f32_16x16x32_f16 %[a1], %[x], %[x], %[a1]\n"
            ".elseif %c[shape] == 2\n"
            "v_mfma_scale_f32_16x16x128_f8f6f4 %[a0], %[y], %[y], %[a0], %[v3], %[v3] op_sel_hi:[0,0,0] cbsz:3 blgp:3\n"
            ".if %c[nv] > 0\n v_add_f32 %[v0], %[v0], %[v1]\n .endif\n"
            ".if %c[nd] > 0\n ds_read_b128 %[d0], %[addr]\n .endif\n"
            ".if %c[nv] > 1\n v_add_f32 %[v1], %[v1], %[v2]\n .endif\n"
            "v_mfma_scale_f32_16x16x128_f8f6f4 %[a1], %[y], %[y], %[a1], %[v3], %[v3] op_sel_hi:[0,0,0] cbsz:3 blgp:3\n"
            ".elseif %c[shape] == 3\n"
            "v_mfma_scale_f32_32x32x64_f8f6f4 %[b0], %[y], %[y], %[b0], %[v3], %[v3] op_sel_hi:[0,0,0] cbsz:3 blgp:3\n"
            ".if %c[nv] > 0\n v_add_f32 %[v0], %[v0], %[v1]\n .endif\n"
            ".if %c[nd] > 0\n ds_read_b128 %[d0], %[addr]\n .endif\n"
            ".if %c[nv] > 1\n v_add_f32 %[v1], %[v1], %[v2]\n .endif\n"
            ".else\n"
            "v_mfma_f32_32x32x16_f16 %[b0], %[x], %[x], %[b0]\n"
            ".if %c[nv] > 0\n v_add_f32 %[v0], %[v0], %[v1]\n .endif\n"
            ".if %c[nd] > 0\n ds_read_b128 %[d0], %[addr]\n .endif\n"
            ".if %c[nv] > 1\n v_add_f32 %[v1], %[v1], %[v2]\n .endif\n"
            ".endif\n"
            ".if %c[nv] > 2\n v_add_f32 %[v2], %[v2], %[v3]\n .endif\n"
            ".if %c[nd] > 1\n ds_read_b128 %[d1], %[addr] offset:4096\n .endif\n"
            ".if %c[nw] > 0\n s_waitcnt lgkmcnt(%c[nd])\n .endif\n"
            ".if %c[nv] > 3\n v_add_f32 %[v3], %[v3], %[v0]\n .endif\n"
            ".if %c[nd] > 2\n ds_read_b128 %[d2], %[addr] offset:8192\n .endif\n"
            ".if %c[nv] > 4\n v_add_f32 %[v0], %[v0], %[v2]\n .endif\n"
            ".if %c[nw] > 1\n s_waitcnt lgkmcnt(%c[nd])\n .endif\n"
            ".if %c[nv] > 5\n v_add_f32 %[v1], %[v1], %[v3]\n .endif\n"
            ".endr\n"
            "s_waitcnt lgkmcnt(0)\n"
            : [a0] "+v"(a0), [a1] "+v"(a1), [b0] "+v"(b0), [v0] "+v"(v0), [v1] "+v"(v1), [v2] "+v"(v2), [v3] "+v"(v3),
              [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2)
            : [x] "v"(x), [y] "v"(y), [addr] "v"(addr), [shape] "i"(SHAPE), [nv] "i"(NV), [nd] "i"(ND), [nw] "i"(NW));
    }
    if (v0 == 12345.f) sink[0] = a0[0] + a1[0] + b0[0] + v1 + v2 + v3 + d0[0] + d1[0] + d2[0];
}

template <int SHAPE, int NV, int ND, int NW>
double run(float* d_sink) {
    const int iters = 4000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<SHAPE, NV, ND, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<SHAPE, NV, ND, NW><<<256, 256, 65536>>>(10, d_sink);
    hipEventRecord(e0);
    k<SHAPE, NV, ND, NW><<<256, 256, 65536>>>(iters, d_sink);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / ((double)iters * 64);   // ns per group
}

#define ROW(NV, ND, NW) { double a = run<0, NV, ND, NW>(d_sink), b = run<1, NV, ND, NW>(d_sink), c = run<2, NV, ND, NW>(d_sink), d = run<3, NV, ND, NW>(d_sink); \
    printf("fillers per group: %d VALU %d ds_read_b128 %d waits: f16 2 x 16x16x32 %.2f ns, 1 x 32x32x16 %.2f ns (%.3f) | bf6 2 x 16x16x128 %.2f ns, 1 x 32x32x64 %.2f ns (%.3f)\n", NV, ND, NW, a, b, b / a, c, d, d / c); }

int main() {
    float* d_sink; hipMalloc(&d_sink, 64);
    ROW(0, 0, 0) ROW(1, 0, 0) ROW(2, 0, 0) ROW(4, 0, 0) ROW(6, 0, 0) ROW(0, 1, 0) ROW(0, 2, 0) ROW(2, 1, 1) ROW(3, 1, 1) ROW(3, 2, 1) ROW(4, 2, 1) ROW(4, 2, 2)
    return 0;
}
