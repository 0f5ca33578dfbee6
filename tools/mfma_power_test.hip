// What does the matrix pipe sustain under the package power limit, and what do the other instruction classes of the
// generated kernels cost in clock?  One wave per SIMD on every CU runs back-to-back MFMAs (4 independent accumulators,
// 8 x 8 operand fragments) for about two seconds per configuration; the operand DATA is zeros, ones or random values
// (the knock-out builds of profiles/r02_cost_structure.txt computed on garbage and ran at clocks the real data does not get).
// Optional fillers per MFMA: ds_read_b128 of random LDS data, v_fma_f32 on random registers.
// Prints for each configuration: effective matrix-pipe clock = MFMA cycles / time (the pipe is never idle by
// construction when there are no fillers), socket power and sclk sampled through rocm_smi while the kernels run.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_power_test.hip -o tools/mfma_power_test -lrocm_smi64 && tools/mfma_power_test
#include <hip/hip_runtime.h>
#include <rocm_smi/rocm_smi.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <chrono>
#include <thread>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x6 __attribute__((ext_vector_type(6)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
extern __shared__ char smem[];

// SHAPE 0: v_mfma_f32_32x32x16_f16 (32 cycles)   1: v_mfma_scale_f32_32x32x64_f8f6f4 bf6 x bf6 (32 cycles)
// 2: v_mfma_f32_16x16x32_f16 (16 cycles)
template <int SHAPE, int ND, int NV, int ORD>
__global__ __launch_bounds__(256, 1) void pw(const i32x4* __restrict__ data, int iters, float* sink) {
    i32x4* l = reinterpret_cast<i32x4*>(smem);
    for (int i = threadIdx.x; i < 4096; i += 256) l[i] = data[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    i32x6 a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const i32x4 p = l[j * 64 + lane], q = l[512 + j * 64 + lane], r = l[1024 + j * 64 + lane];
        a[j] = i32x6{p[0], p[1], p[2], p[3], r[0], r[1]};
        b[j] = i32x6{q[0], q[1], q[2], q[3], r[2], r[3]};
    }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    f32x4 c4 = {0, 0, 0, 0}, c5 = c4, c6 = c4, c7 = c4;
    int sc = 127;
    f32x4 d0, d1;
    float v0 = __int_as_float(l[2048 + lane][0] & 0x3fffffff), v1 = __int_as_float(l[2100 + lane][1] & 0x3fffffff), v2 = 0.f, v3 = 0.f;
    unsigned addr = lane * 16;
#define MF(c, x, y)                                                                                                     \
    ".if %c[shape] == 0\n v_mfma_f32_32x32x16_f16 %[" #c "], %[" #x "], %[" #y "], %[" #c "]\n"                          \
    ".elseif %c[shape] == 1\n v_mfma_scale_f32_32x32x64_f8f6f4 %[" #c "], %[" #x "x], %[" #y "x], %[" #c "], %[sc], %[sc] op_sel_hi:[0,0,0] cbsz:3 blgp:3\n" \
    ".else\n v_mfma_f32_16x16x32_f16 %[" #c "s], %[" #x "], %[" #y "], %[" #c "s]\n .endif\n"                            \
    ".if %c[nd] > 0\n ds_read_b128 %[d0], %[addr] offset:(1024 * (\\@ & 31))\n .endif\n"                                  \
    ".if %c[nv] > 0\n v_fma_f32 %[v2], %[v0], %[v1], %[v2]\n .endif\n"                                                    \
    ".if %c[nd] > 1\n ds_read_b128 %[d1], %[addr] offset:(1024 * ((\\@ + 7) & 31) + 32768)\n .endif\n"                    \
    ".if %c[nv] > 1\n v_fma_f32 %[v3], %[v1], %[v0], %[v3]\n .endif\n"                                                    \
    ".if %c[nv] > 2\n v_fma_f32 %[v2], %[v0], %[v1], %[v2]\n .endif\n"                                                    \
    ".if %c[nv] > 3\n v_fma_f32 %[v3], %[v1], %[v0], %[v3]\n .endif\n"
    for (int i = 0; i < iters; ++i) {
        asm volatile(
            ".macro GRP\n"
            ".if %c[ord] == 0\n"   // no operand shared between consecutive MFMAs
            MF(c0, a0, b0) MF(c1, a1, b1) MF(c2, a2, b2) MF(c3, a3, b3)
            MF(c0, a4, b5) MF(c1, a5, b6) MF(c2, a6, b7) MF(c3, a7, b4)
            MF(c0, a1, b2) MF(c1, a2, b3) MF(c2, a3, b0) MF(c3, a0, b1)
            MF(c0, a6, b4) MF(c1, a7, b5) MF(c2, a4, b6) MF(c3, a5, b7)
            ".elseif %c[ord] == 1\n"   // pairs share B
            MF(c0, a0, b0) MF(c1, a1, b0) MF(c2, a2, b1) MF(c3, a3, b1)
            MF(c0, a4, b2) MF(c1, a5, b2) MF(c2, a6, b3) MF(c3, a7, b3)
            MF(c0, a1, b4) MF(c1, a2, b4) MF(c2, a3, b5) MF(c3, a0, b5)
            MF(c0, a6, b6) MF(c1, a7, b6) MF(c2, a4, b7) MF(c3, a5, b7)
            ".elseif %c[ord] == 2\n"   // pairs share A
            MF(c0, a0, b0) MF(c1, a0, b1) MF(c2, a1, b2) MF(c3, a1, b3)
            MF(c0, a2, b4) MF(c1, a2, b5) MF(c2, a3, b6) MF(c3, a3, b7)
            MF(c0, a4, b1) MF(c1, a4, b2) MF(c2, a5, b3) MF(c3, a5, b0)
            MF(c0, a6, b5) MF(c1, a6, b6) MF(c2, a7, b7) MF(c3, a7, b4)
            ".elseif %c[ord] == 3\n"   // quads share B
            MF(c0, a0, b0) MF(c1, a1, b0) MF(c2, a2, b0) MF(c3, a3, b0)
            MF(c0, a4, b1) MF(c1, a5, b1) MF(c2, a6, b1) MF(c3, a7, b1)
            MF(c0, a1, b2) MF(c1, a2, b2) MF(c2, a3, b2) MF(c3, a0, b2)
            MF(c0, a6, b3) MF(c1, a7, b3) MF(c2, a4, b3) MF(c3, a5, b3)
            ".else\n"                  // the same two operands every time
            MF(c0, a0, b0) MF(c1, a0, b0) MF(c2, a0, b0) MF(c3, a0, b0)
            MF(c0, a0, b0) MF(c1, a0, b0) MF(c2, a0, b0) MF(c3, a0, b0)
            MF(c0, a0, b0) MF(c1, a0, b0) MF(c2, a0, b0) MF(c3, a0, b0)
            MF(c0, a0, b0) MF(c1, a0, b0) MF(c2, a0, b0) MF(c3, a0, b0)
            ".endif\n"
            ".endm\n"
            "GRP\n GRP\n GRP\n GRP\n"
            ".purgem GRP\n"
            "s_waitcnt lgkmcnt(0)\n"
            : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [c0s] "+v"(c4), [c1s] "+v"(c5), [c2s] "+v"(c6),
              [c3s] "+v"(c7), [v2] "+v"(v2), [v3] "+v"(v3), [d0] "=&v"(d0), [d1] "=&v"(d1)
            : [a0] "v"(__builtin_shufflevector(a[0], a[0], 0, 1, 2, 3)), [a1] "v"(__builtin_shufflevector(a[1], a[1], 0, 1, 2, 3)),
              [a2] "v"(__builtin_shufflevector(a[2], a[2], 0, 1, 2, 3)), [a3] "v"(__builtin_shufflevector(a[3], a[3], 0, 1, 2, 3)),
              [a4] "v"(__builtin_shufflevector(a[4], a[4], 0, 1, 2, 3)), [a5] "v"(__builtin_shufflevector(a[5], a[5], 0, 1, 2, 3)),
              [a6] "v"(__builtin_shufflevector(a[6], a[6], 0, 1, 2, 3)), [a7] "v"(__builtin_shufflevector(a[7], a[7], 0, 1, 2, 3)),
              [b0] "v"(__builtin_shufflevector(b[0], b[0], 0, 1, 2, 3)), [b1] "v"(__builtin_shufflevector(b[1], b[1], 0, 1, 2, 3)),
              [b2] "v"(__builtin_shufflevector(b[2], b[2], 0, 1, 2, 3)), [b3] "v"(__builtin_shufflevector(b[3], b[3], 0, 1, 2, 3)),
              [b4] "v"(__builtin_shufflevector(b[4], b[4], 0, 1, 2, 3)), [b5] "v"(__builtin_shufflevector(b[5], b[5], 0, 1, 2, 3)),
              [b6] "v"(__builtin_shufflevector(b[6], b[6], 0, 1, 2, 3)), [b7] "v"(__builtin_shufflevector(b[7], b[7], 0, 1, 2, 3)),
              [a0x] "v"(a[0]), [a1x] "v"(a[1]), [a2x] "v"(a[2]), [a3x] "v"(a[3]), [a4x] "v"(a[4]), [a5x] "v"(a[5]),
              [a6x] "v"(a[6]), [a7x] "v"(a[7]), [b0x] "v"(b[0]), [b1x] "v"(b[1]), [b2x] "v"(b[2]), [b3x] "v"(b[3]),
              [b4x] "v"(b[4]), [b5x] "v"(b[5]), [b6x] "v"(b[6]), [b7x] "v"(b[7]),
              [sc] "v"(sc), [v0] "v"(v0), [v1] "v"(v1), [addr] "v"(addr), [shape] "i"(SHAPE), [nd] "i"(ND), [nv] "i"(NV), [ord] "i"(ORD));
    }
    float s = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3] + v2 + v3 + d0[0] + d1[1];
    if (s == 12345.678f) sink[0] = s;
}

static uint64_t power_uw() {
    uint64_t p = 0;
    if (rsmi_dev_current_socket_power_get(0, &p) != RSMI_STATUS_SUCCESS) {
        RSMI_POWER_TYPE t;
        if (rsmi_dev_power_get(0, &p, &t) != RSMI_STATUS_SUCCESS) p = 0;
    }
    return p;
}
static double sclk_mhz() {
    rsmi_frequencies_t f;
    if (rsmi_dev_gpu_clk_freq_get(0, RSMI_CLK_TYPE_SYS, &f) != RSMI_STATUS_SUCCESS) return 0;
    return f.frequency[f.current] / 1e6;
}

template <int SHAPE, int ND, int NV, int ORD = 0>
void run(const char* what, const char* dname, const i32x4* d_data, float* d_sink) {
    const void* fn = reinterpret_cast<const void*>(&pw<SHAPE, ND, NV, ORD>);
    hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const int iters = 2000;   // x 64 MFMAs x 32 (16) cycles = 4.1 (2.0) M cycles: about 2 ms per launch
    const double cyc = (double)iters * 64 * (SHAPE == 2 ? 16 : 32);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    pw<SHAPE, ND, NV, ORD><<<256, 256, 65536>>>(d_data, 10, d_sink);
    hipDeviceSynchronize();
    const int launches = 1000;
    hipEventRecord(e0);
    for (int i = 0; i < launches - 300; ++i) pw<SHAPE, ND, NV, ORD><<<256, 256, 65536>>>(d_data, iters, d_sink);
    hipEventRecord(e0);   // the last 300 launches are the steady-state window
    for (int i = 0; i < 300; ++i) pw<SHAPE, ND, NV, ORD><<<256, 256, 65536>>>(d_data, iters, d_sink);
    hipEventRecord(e1);
    double pw_sum = 0, ck_sum = 0; int n = 0;
    while (hipEventQuery(e1) == hipErrorNotReady) {
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
        if (hipEventQuery(e0) == hipErrorNotReady) continue;
        pw_sum += power_uw() * 1e-6; ck_sum += sclk_mhz(); ++n;
    }
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ghz = cyc * 300 / (ms * 1e-3) * 1e-9;
    printf("%-34s %-8s fillers/MFMA: %d ds_read_b128 %d v_fma  order %d   pipe cycles/s %.3f GHz   power %.0f W  sclk %.0f MHz (%d samples)\n",
           what, dname, ND, NV, ORD, ghz, n ? pw_sum / n : 0.0, n ? ck_sum / n : 0.0, n);
    fflush(stdout);
}

static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }

int main() {
    rsmi_init(0);
    float* d_sink; hipMalloc(&d_sink, 64);
    i32x4* d_data; hipMalloc(&d_data, 65536);
    std::vector<uint16_t> h(32768);
    srand(1);
    auto gauss = []() { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return sqrt(-2 * log(u)) * cos(6.283185307 * v); };
    for (int mode = 0; mode < 4; ++mode) {
        const char* dname = mode == 0 ? "zeros" : mode == 1 ? "ones" : mode == 2 ? "N(0,1)" : "relu";
        for (auto& x : h) {
            const double g = gauss();
            x = mode == 0 ? 0 : mode == 1 ? f2h(1.f) : mode == 2 ? f2h((float)g) : f2h(g > 0 ? (float)g : 0.f);
        }
        hipMemcpy(d_data, h.data(), 65536, hipMemcpyHostToDevice);
        run<0, 0, 0>("v_mfma_f32_32x32x16_f16", dname, d_data, d_sink);
        run<2, 0, 0>("v_mfma_f32_16x16x32_f16", dname, d_data, d_sink);
        if (mode == 1) continue;
        // for the bf6 shape the same bits are read as 6-bit values: zeros, or every pattern about equally often
        run<1, 0, 0>("v_mfma_scale_f32_32x32x64 bf6", dname, d_data, d_sink);
        if (mode == 2) {
            // operand order: 1 pairs share B, 2 pairs share A, 3 quads share B, 4 one operand pair throughout
            run<0, 0, 0, 1>("v_mfma_f32_32x32x16_f16", dname, d_data, d_sink);
            run<0, 0, 0, 2>("v_mfma_f32_32x32x16_f16", dname, d_data, d_sink);
            run<0, 0, 0, 3>("v_mfma_f32_32x32x16_f16", dname, d_data, d_sink);
            run<0, 0, 0, 4>("v_mfma_f32_32x32x16_f16", dname, d_data, d_sink);
            run<2, 0, 0, 1>("v_mfma_f32_16x16x32_f16", dname, d_data, d_sink);
            run<2, 0, 0, 4>("v_mfma_f32_16x16x32_f16", dname, d_data, d_sink);
            run<1, 0, 0, 1>("v_mfma_scale_f32_32x32x64 bf6", dname, d_data, d_sink);
            run<1, 0, 0, 4>("v_mfma_scale_f32_32x32x64 bf6", dname, d_data, d_sink);
            run<0, 1, 0>("v_mfma_f32_32x32x16_f16", dname, d_data, d_sink);
            run<0, 2, 0>("v_mfma_f32_32x32x16_f16", dname, d_data, d_sink);
            run<0, 0, 2>("v_mfma_f32_32x32x16_f16", dname, d_data, d_sink);
            run<0, 0, 4>("v_mfma_f32_32x32x16_f16", dname, d_data, d_sink);
            run<0, 1, 2>("v_mfma_f32_32x32x16_f16", dname, d_data, d_sink);
            run<1, 1, 0>("v_mfma_scale_f32_32x32x64 bf6", dname, d_data, d_sink);
            run<1, 2, 2>("v_mfma_scale_f32_32x32x64 bf6", dname, d_data, d_sink);
        }
    }
    rsmi_shut_down();
    return 0;
}
