// Energy account of the generated kernels: what does each instruction class of r2l_body_kernel /
// nerf_chain_kernel cost in WATTS?  Both kernels run at the package power limit, so the currency is joules per ray, not
// cycles.  One workgroup of 4 waves per CU (256 CUs), each wave loops over a body of 16 "slots"; per configuration the
// slots hold a chosen number of each class, spread evenly (Bresenham) over the 16 slots:
//   M   v_mfma_f32_32x32x16_f16            (32 pipe cycles)       A = N(0, 1) weights, B = relu(N(0, 1)) activations
//   H   v_mfma_f32_16x16x32_f16            (16 pipe cycles)
//   B   v_mfma_scale_f32_32x32x64_f8f6f4   bf6 x bf6 (32 pipe cycles), random 6-bit patterns
//   R   ds_read_b128                       (1 KiB per wave instruction)
//   W   ds_write_b128
//   G   global_load_lds_dwordx4            (LDS-DMA, 1 KiB per wave instruction, from a 19 MiB buffer: L2 / MALL resident)
//   L   global_load_dwordx4                (the same bytes into VGPRs)
//   C   v_cvt_scalef32_pk32_bf6_f16        (32 values per lane)
//   X   v_fma_mixlo_f16 + v_fma_mixhi_f16  (one pair)
//   P   v_cvt_pk_f16_f32
//   F   v_fma_f32
//   A   v_accvgpr_write_b32 + v_accvgpr_read_b32 (one pair)
//   E   v_mfma_scale_f32_32x32x64_f8f6f4   e4m3 x e4m3 (64 pipe cycles), random bytes            (round 5: r2l_body8_kernel's terms)
//   Q   v_cvt_scalef32_pk_fp8_f16          (2 values per lane)
// For every configuration: iterations / s (-> events / s per class), socket power and sclk through rocm_smi over a
// ~2.5 s steady-state window.  tools/energy_fit.py turns the table into joules per event and prices the two kernels'
// instruction mixes at their measured rates (profiles/r04_energy_account.txt).
//   hipcc --offload-arch=gfx950 -O3 tools/energy_probe.hip -o tools/energy_probe -lrocm_smi64 && tools/energy_probe
#include <hip/hip_runtime.h>
#include <rocm_smi/rocm_smi.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x6 __attribute__((ext_vector_type(6)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
extern __shared__ char smem[];

#define STREAM_BYTES (19u << 20)

struct Cfg {
    int M, H, B, R, W, G, L, C, X, P, F, A;
};

// slot s of 16 holds floor((s + 1) n / 16) - floor(s n / 16) instructions of a class with n per iteration
#define SLOTS(n, text) ".rept ((eps_s + 1) * %c[" #n "]) / 16 - (eps_s * %c[" #n "]) / 16\n" text ".endr\n"

template <int M, int H, int B, int R, int W, int G, int L, int C, int X, int P, int F, int A, int K, int LS, int E8, int Q8>
__global__ __launch_bounds__(256, 1) void probe(const i32x4* __restrict__ data, const char* __restrict__ stream, int iters, float* sink) {
    i32x4* l = reinterpret_cast<i32x4*>(smem);
    for (int i = threadIdx.x; i < 4096; i += 256) l[i] = data[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    i32x6 a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const i32x4 p = l[j * 64 + lane], q = l[2048 + j * 64 + lane], r = l[1024 + j * 64 + lane];
        a[j] = i32x6{p[0], p[1], p[2], p[3], r[0], r[1]};     // weights: N(0, 1)
        b[j] = i32x6{q[0], q[1], q[2], q[3], r[2], r[3]};     // activations: relu(N(0, 1))
    }
    const i32x8 a8 = {a[0][0], a[0][1], a[0][2], a[0][3], a[1][0], a[1][1], a[1][2], a[1][3]};     // 32 random bytes per lane as e4m3 operands
    const i32x8 b8 = {b[0][0], b[0][1], b[0][2], b[0][3], b[1][0], b[1][1], b[1][2], b[1][3]};
    unsigned q8 = 0;
    f32x16 c0 = {0}, c1 = {0};
    f32x4 h0 = {0, 0, 0, 0}, h1 = h0;
    int sc = 127;
    f32x4 d0 = {0, 0, 0, 0}, g0 = d0;
    f32x4 d1 = __builtin_bit_cast(f32x4, l[1500 + lane]);     // ds_write data: random bits
    i32x16 src;
#pragma unroll
    for (int j = 0; j < 16; ++j) src[j] = l[2200 + 64 * (j >> 2) + lane][j & 3];   // 32 f16 activations per lane
    i32x16 src2 = src;
    i32x6 cv = {0, 0, 0, 0, 0, 0};
    float v0 = __int_as_float(l[2048 + lane][0] & 0x3fffffff), v1 = __int_as_float(l[2100 + lane][1] & 0x3fffffff), v2 = 0.f, v3 = 0.f;
    float scale = 1.0f;
    unsigned x0 = 0, x1 = 0, acc_t = 0;
    unsigned addr = lane * 16;                 // ds_read / ds_write: rotating 1 KiB pieces of a 32 KiB window per wave
    unsigned waddr = 65536 + wave * 8192 + lane * 16;
    // the stream: every wave walks the 19 MiB buffer in 1 KiB steps from its own offset (blocks start spread over it)
    const char* gbase = stream;
    // LS = 0: every wave walks its own part of the buffer (no two CUs share a line: the stream comes from the Infinity Cache);
    // LS = 1: all workgroups walk the SAME addresses in step, wave w taking every 4th KiB -- the generated kernels' pattern
    // (every CU streams the same weights at about the same time: L2 hits behind the first CU of an XCD)
    unsigned goff = LS ? (unsigned)(wave * 1024 + lane * 16)
                       : (unsigned)(((blockIdx.x * 4 + wave) * 73 * 1024) % (STREAM_BYTES - (1u << 20))) + lane * 16;
    const unsigned lds_dma_base = __builtin_amdgcn_readfirstlane(98304 + wave * 4096);    // M0: where the wave's LDS-DMA pieces land (4 KiB window per wave)
    for (int i = 0; i < iters; ++i) {
        asm volatile(
            "s_mov_b32 m0, %[m0v]\n"
            "s_nop 0\n"
            ".set eps_s, 0\n"
            ".rept 16\n"
            SLOTS(nM, ".if (eps_s & 1)\n v_mfma_f32_32x32x16_f16 %[c1], %[a1], %[b1], %[c1]\n .else\n v_mfma_f32_32x32x16_f16 %[c0], %[a0], %[b0], %[c0]\n .endif\n")
            SLOTS(nG, "global_load_lds_dwordx4 %[goff], %[gbase]\n v_add_u32 %[goff], %[gstep], %[goff]\n")
            SLOTS(nR, "ds_read_b128 %[d0], %[addr] offset:(1024 * (eps_s & 15))\n")
            SLOTS(nF, "v_fma_f32 %[v2], %[v0], %[v1], %[v2]\n")
            SLOTS(nH, ".if (eps_s & 1)\n v_mfma_f32_16x16x32_f16 %[h1], %[a3], %[b3], %[h1]\n .else\n v_mfma_f32_16x16x32_f16 %[h0], %[a2], %[b2], %[h0]\n .endif\n")
            SLOTS(nP, "v_cvt_pk_f16_f32 %[x0], %[v0], %[v1]\n")
            SLOTS(nB, ".if (eps_s & 1)\n v_mfma_scale_f32_32x32x64_f8f6f4 %[c1], %[a1x], %[b1x], %[c1], %[sc], %[sc] op_sel_hi:[0,0,0] cbsz:3 blgp:3\n .else\n"
                      " v_mfma_scale_f32_32x32x64_f8f6f4 %[c0], %[a0x], %[b0x], %[c0], %[sc], %[sc] op_sel_hi:[0,0,0] cbsz:3 blgp:3\n .endif\n")
            SLOTS(nL, "global_load_dwordx4 %[g0], %[goff], %[gbase]\n v_add_u32 %[goff], %[gstep], %[goff]\n")
            SLOTS(nK, ".if (eps_s & 1)\n v_mfma_scale_f32_16x16x128_f8f6f4 %[h1], %[a1x], %[b1x], %[h1], %[sc], %[sc] op_sel_hi:[0,0,0] cbsz:3 blgp:3\n .else\n"
                      " v_mfma_scale_f32_16x16x128_f8f6f4 %[h0], %[a0x], %[b0x], %[h0], %[sc], %[sc] op_sel_hi:[0,0,0] cbsz:3 blgp:3\n .endif\n")
            SLOTS(nW, "ds_write_b128 %[waddr], %[d1] offset:(1024 * (eps_s & 7))\n")
            SLOTS(nX, "v_fma_mixlo_f16 %[x1], %[x0], %[scale], %[v0] op_sel_hi:[1,0,0]\n v_fma_mixhi_f16 %[x1], %[x0], %[scale], %[v1] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n")
            SLOTS(nC, ".if (eps_s & 1)\n v_cvt_scalef32_pk32_bf6_f16 %[cv], %[src2], %[scale]\n .else\n v_cvt_scalef32_pk32_bf6_f16 %[cv], %[src], %[scale]\n .endif\n")
            SLOTS(nA, "v_accvgpr_write_b32 a0, %[v0]\n s_nop 0\n v_accvgpr_read_b32 %[acct], a0\n")
            SLOTS(nE, ".if (eps_s & 1)\n v_mfma_scale_f32_32x32x64_f8f6f4 %[c1], %[a8], %[b8], %[c1], %[sc], %[sc] op_sel_hi:[0,0,0]\n .else\n"
                      " v_mfma_scale_f32_32x32x64_f8f6f4 %[c0], %[b8], %[a8], %[c0], %[sc], %[sc] op_sel_hi:[0,0,0]\n .endif\n")
            SLOTS(nQ, "v_cvt_scalef32_pk_fp8_f16 %[q8], %[x0], %[scale]\n")
            ".set eps_s, eps_s + 1\n"
            ".endr\n"
            // wrap the stream offset (scalar compare on lane 0's value is not needed: every lane adds the same steps)
            "v_cmp_lt_u32 vcc, %[wrap], %[goff]\n"
            "v_subrev_u32 %[goff], %[wrap], %[goff]\n"
            "v_add_u32 %[goff], %[wrap], %[goff]\n"
            "v_cndmask_b32 %[goff], %[goff], %[start], vcc\n"
            // the loads of THIS iteration stay in flight across the loop edge (the generated kernels prefetch chunks ahead): only
            // the previous iteration's are waited for
            "s_waitcnt vmcnt(%c[nG] + %c[nL]) lgkmcnt(0)\n"
            : [c0] "+v"(c0), [c1] "+v"(c1), [h0] "+v"(h0), [h1] "+v"(h1), [v2] "+v"(v2), [d0] "+v"(d0), [g0] "+v"(g0), [goff] "+v"(goff),
              [x0] "+v"(x0), [x1] "+v"(x1), [cv] "+v"(cv), [acct] "+v"(acc_t), [q8] "+v"(q8)
            : [a0] "v"(__builtin_shufflevector(a[0], a[0], 0, 1, 2, 3)), [a1] "v"(__builtin_shufflevector(a[1], a[1], 0, 1, 2, 3)),
              [a2] "v"(__builtin_shufflevector(a[2], a[2], 0, 1, 2, 3)), [a3] "v"(__builtin_shufflevector(a[3], a[3], 0, 1, 2, 3)),
              [b0] "v"(__builtin_shufflevector(b[0], b[0], 0, 1, 2, 3)), [b1] "v"(__builtin_shufflevector(b[1], b[1], 0, 1, 2, 3)),
              [b2] "v"(__builtin_shufflevector(b[2], b[2], 0, 1, 2, 3)), [b3] "v"(__builtin_shufflevector(b[3], b[3], 0, 1, 2, 3)),
              [a0x] "v"(a[0]), [a1x] "v"(a[1]), [b0x] "v"(b[0]), [b1x] "v"(b[1]), [a8] "v"(a8), [b8] "v"(b8), [sc] "v"(sc), [v0] "v"(v0), [v1] "v"(v1),
              [addr] "v"(addr), [waddr] "v"(waddr), [d1] "v"(d1), [gbase] "s"(gbase), [src] "v"(src), [src2] "v"(src2), [scale] "v"(scale),
              [m0v] "s"(lds_dma_base), [wrap] "v"(STREAM_BYTES - (2u << 20)), [start] "v"(LS ? (unsigned)(wave * 1024 + lane * 16) : (unsigned)lane * 16), [gstep] "v"(LS ? 4096u : 1024u), [nK] "i"(K),
              [nM] "i"(M), [nH] "i"(H), [nB] "i"(B), [nR] "i"(R), [nW] "i"(W), [nG] "i"(G), [nL] "i"(L), [nC] "i"(C), [nX] "i"(X),
              [nP] "i"(P), [nF] "i"(F), [nA] "i"(A), [nE] "i"(E8), [nQ] "i"(Q8)
            : "a0", "vcc", "memory");
    }
    float s = c0[0] + c1[1] + h0[2] + h1[3] + v2 + v3 + d0[0] + g0[1] + __int_as_float(x0 ^ x1 ^ acc_t ^ cv[0] ^ cv[5] ^ q8);
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

static const i32x4* g_data;
static const char* g_stream;
static float* g_sink;

template <int M, int H, int B, int R, int W, int G, int L, int C, int X, int P, int F, int A, int K = 0, int LS = 0, int E8 = 0, int Q8 = 0>
void run(const char* tag) {
    auto kern = probe<M, H, B, R, W, G, L, C, X, P, F, A, K, LS, E8, Q8>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    // calibrate the iteration count to ~2.5 ms per launch
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    kern<<<256, 256, 131072>>>(g_data, g_stream, 8, g_sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<256, 256, 131072>>>(g_data, g_stream, 200, g_sink);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    int iters = (int)(200 * 2.5 / (ms > 1e-3 ? ms : 1e-3));
    if (iters < 50) iters = 50;
    const int warm = 300, window = 800;
    for (int i = 0; i < warm; ++i) kern<<<256, 256, 131072>>>(g_data, g_stream, iters, g_sink);
    hipEventRecord(e0);
    for (int i = 0; i < window; ++i) kern<<<256, 256, 131072>>>(g_data, g_stream, iters, g_sink);
    hipEventRecord(e1);
    double pw_sum = 0, ck_sum = 0;
    int n = 0;
    while (hipEventQuery(e1) == hipErrorNotReady) {
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
        if (hipEventQuery(e0) == hipErrorNotReady) continue;
        pw_sum += power_uw() * 1e-6;
        ck_sum += sclk_mhz();
        ++n;
    }
    hipDeviceSynchronize();
    hipError_t err = hipGetLastError();
    hipEventElapsedTime(&ms, e0, e1);
    const double it_per_s = (double)iters * window / (ms * 1e-3);        // per wave; x 1024 waves on the chip
    printf("%-34s M %2d H %2d B %2d K %2d E %2d Q %2d R %2d W %2d G %2d L %2d%s C %2d X %2d P %2d F %2d A %2d   iter/s/wave %.4e   power %7.1f W   sclk %6.0f MHz   (%d samples)%s\n",
           tag, M, H, B, K, E8, Q8, R, W, G, L, LS ? " lockstep" : "         ", C, X, P, F, A, it_per_s, n ? pw_sum / n : 0.0, n ? ck_sum / n : 0.0, n,
           err == hipSuccess ? "" : hipGetErrorString(err));
    fflush(stdout);
}

static uint16_t f2h(float f) {
    _Float16 h = (_Float16)f;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}

int main() {
    rsmi_init(0);
    float* d_sink;
    hipMalloc(&d_sink, 64);
    i32x4* d_data;
    hipMalloc(&d_data, 65536);
    char* d_stream;
    hipMalloc(&d_stream, STREAM_BYTES);
    std::vector<uint16_t> h(32768);
    srand(1);
    auto gauss = []() {
        double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
        return sqrt(-2 * log(u)) * cos(6.283185307 * v);
    };
    // l[0 .. 1023] (i32x4 units): weights N(0, 1); [1024 .. 2047]: random bits; [2048 ..]: relu(N(0, 1)) activations
    for (size_t i = 0; i < h.size(); ++i) {
        const double g = gauss();
        h[i] = i < 8192 ? f2h((float)g) : i < 16384 ? (uint16_t)(rand() & 0xffff) : f2h(g > 0 ? (float)g : 0.f);
    }
    hipMemcpy(d_data, h.data(), 65536, hipMemcpyHostToDevice);
    std::vector<uint16_t> st(STREAM_BYTES / 2);
    for (auto& x : st) x = f2h((float)gauss());
    hipMemcpy(d_stream, st.data(), STREAM_BYTES, hipMemcpyHostToDevice);
    g_data = d_data;
    g_stream = d_stream;
    g_sink = d_sink;
    {   // idle: no kernel for 2 s
        double p = 0, c = 0;
        for (int i = 0; i < 20; ++i) {
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
            p += power_uw() * 1e-6;
            c += sclk_mhz();
        }
        printf("%-28s power %7.1f W   sclk %6.0f MHz\n", "idle (no kernel)", p / 20, c / 20);
    }
    //   M  H  B  R  W  G  L  C  X  P  F  A [K LS]
    if (getenv("EP_E4M3") != nullptr) {
        // round 5, second part: the kernel the trained-like student lands on since the split rungs, r2l_body8_kernel
        // (e4m3 terms).  Per block and wave: 256 fp16 MFMAs (32x32x16) + 128 e4m3 MFMAs (32x32x64, 64 pipe cycles each), 544 KiB of ds_read,
        // 129 LDS-DMA, 256 v_cvt_scalef32_pk_fp8_f16, 128 v_fma_mix pairs, 128 v_cvt_pk, 128 v_accvgpr pairs, ~130 plain VALU
        // -> per 16 MFMAs: M 11, E 5, R 23, G 5, Q 11, X 5, P 5, F 5, A 5
        run<0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 16, 0>("mfma 32x32x64 e4m3 x16");
        run<0, 0, 16, 0, 0, 0, 0, 0, 0, 0, 0, 0>("mfma 32x32x64 bf6 x16");
        run<16, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0>("mfma 32x32x16 f16 x16");
        run<0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 64>("v_cvt_scalef32_pk_fp8_f16 x64");
        run<11, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 5, 0>("mfma 32x32 f16:e4m3 2:1");
        run<11, 0, 5, 0, 0, 0, 0, 0, 0, 0, 0, 0>("mfma 32x32 f16:bf6 2:1");
        run<11, 0, 0, 23, 0, 5, 0, 0, 5, 5, 5, 5, 0, 1, 5, 11>("body8 replica (e4m3 terms)");
        run<11, 0, 5, 22, 0, 5, 0, 1, 5, 5, 8, 4, 0, 1>("body replica (bf6 terms)");
        run<11, 0, 0, 23, 0, 0, 0, 0, 5, 5, 5, 5, 0, 1, 5, 11>("body8 replica, no DMA");
        run<11, 0, 0, 0, 0, 5, 0, 0, 5, 5, 5, 5, 0, 1, 5, 11>("body8 replica, no ds_read");
        run<11, 0, 0, 23, 0, 5, 0, 0, 0, 0, 0, 0, 0, 1, 5, 0>("body8 replica, no VALU");
        run<11, 0, 0, 23, 0, 5, 0, 0, 5, 5, 5, 5, 0, 1, 5, 11>("body8 replica (e4m3 terms) again");
        rsmi_shut_down();
        return 0;
    }
    if (getenv("EP_X3") != nullptr) {
        // round 5: the mix of the kernels trained-like weights end on, in both MFMA shapes.  r2l_bodyx_kernel per
        // block and wave: 768 fp16 MFMAs (32x32x16), 544 KiB of ds_read, 129 LDS-DMA, 128 v_fma_mix pairs, 128 v_cvt_pk, 192 v_accvgpr
        // pairs, 385 plain VALU -> per 16 MFMAs: R 11, G 3, X 3, P 3, F 8, A 4.  In 16x16x32 shapes the same MACs are 32 MFMAs and, with
        // the wave's 32 rays as two column tiles per A fragment, the same LDS bytes.  The teacher's three-pass chain (16x16 today): per
        // 32 MFMAs R 11, G 3, X 6, F 5, A 1.
        run<16, 0, 0, 11, 0, 3, 0, 0, 3, 3, 8, 4, 0, 1>("bodyx replica (32x32 shapes)");
        run<0, 32, 0, 11, 0, 3, 0, 0, 3, 3, 8, 4, 0, 1>("bodyx replica in 16x16 shapes");
        run<16, 0, 0, 11, 0, 0, 0, 0, 3, 3, 8, 4, 0, 1>("bodyx replica (32x32), no DMA");
        run<0, 32, 0, 11, 0, 0, 0, 0, 3, 3, 8, 4, 0, 1>("bodyx replica (16x16), no DMA");
        run<16, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0>("mfma 32x32x16 f16 x16");
        run<0, 32, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0>("mfma 16x16x32 f16 x32");
        run<0, 32, 0, 11, 0, 3, 0, 0, 6, 0, 5, 1, 0, 1>("three-pass chain replica (16x16)");
        run<16, 0, 0, 11, 0, 3, 0, 0, 6, 0, 5, 1, 0, 1>("three-pass chain replica in 32x32 shapes");
        run<16, 0, 0, 11, 0, 3, 0, 0, 3, 3, 8, 4, 0, 1>("bodyx replica (32x32 shapes) again");
        run<0, 32, 0, 11, 0, 3, 0, 0, 3, 3, 8, 4, 0, 1>("bodyx replica in 16x16 shapes again");
        rsmi_shut_down();
        return 0;
    }
    if (getenv("EP_AB") != nullptr) {
        // A/B of the energy account's first lever (profiles/r04_energy_account.txt): the body kernel's mix with the SAME bytes,
        // VALU and MACs in 32x32 and in 16x16 MFMA shapes, loads in flight across the loop edge (power-bound, not latency-bound)
        run<11, 0, 5, 22, 0, 5, 0, 1, 5, 5, 8, 4, 0, 1>("body replica (32x32 shapes)");
        run<0, 22, 0, 22, 0, 5, 0, 1, 5, 5, 8, 4, 10, 1>("body replica in 16x16 shapes");
        run<0, 22, 5, 22, 0, 5, 0, 1, 5, 5, 8, 4, 0, 1>("body replica, f16 16x16 + bf6 32x32");
        run<11, 0, 5, 22, 0, 0, 0, 1, 5, 5, 8, 4, 0, 1>("body replica (32x32), no DMA");
        run<0, 22, 0, 22, 0, 0, 0, 1, 5, 5, 8, 4, 10, 1>("body replica (16x16), no DMA");
        run<11, 0, 5, 11, 3, 0, 5, 1, 5, 5, 8, 4, 0, 1>("B-from-LDS variant (32x32)");
        run<0, 22, 0, 11, 3, 0, 5, 1, 5, 5, 8, 4, 10, 1>("B-from-LDS variant (16x16)");
        run<0, 22, 0, 20, 0, 4, 0, 1, 8, 8, 10, 3, 10, 1>("chain replica (16x16)");
        run<0, 22, 0, 8, 3, 0, 4, 1, 8, 8, 10, 3, 10, 1>("chain, weights in registers");
        run<11, 0, 5, 22, 0, 5, 0, 1, 5, 5, 8, 4, 0, 1>("body replica (32x32 shapes) again");
        run<0, 22, 0, 22, 0, 5, 0, 1, 5, 5, 8, 4, 10, 1>("body replica in 16x16 shapes again");
        rsmi_shut_down();
        return 0;
    }
    run<0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0>("spin (loop overhead only)");
    run<0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 64, 0>("v_fma_f32 x64");
    run<0, 0, 0, 0, 0, 0, 0, 0, 0, 32, 0, 0>("v_cvt_pk_f16_f32 x32");
    run<0, 0, 0, 0, 0, 0, 0, 0, 32, 0, 0, 0>("v_fma_mix pair x32");
    run<0, 0, 0, 0, 0, 0, 0, 8, 0, 0, 0, 0>("cvt_pk32_bf6 x8");
    run<0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 32>("accvgpr pair x32");
    run<0, 0, 0, 16, 0, 0, 0, 0, 0, 0, 0, 0>("ds_read_b128 x16");
    run<0, 0, 0, 0, 16, 0, 0, 0, 0, 0, 0, 0>("ds_write_b128 x16 (random data)");
    run<0, 0, 0, 0, 0, 16, 0, 0, 0, 0, 0, 0>("LDS-DMA x16 (Infinity Cache)");
    run<0, 0, 0, 0, 0, 16, 0, 0, 0, 0, 0, 0, 0, 1>("LDS-DMA x16 (L2, lockstep)");
    run<0, 0, 0, 0, 0, 4, 0, 0, 0, 0, 0, 0, 0, 1>("LDS-DMA x4 (L2, lockstep)");
    run<0, 0, 0, 0, 0, 0, 16, 0, 0, 0, 0, 0, 0, 1>("global_load x16 (L2, lockstep)");
    run<16, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0>("mfma 32x32x16 f16 x16");
    run<0, 32, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0>("mfma 16x16x32 f16 x32");
    run<0, 0, 16, 0, 0, 0, 0, 0, 0, 0, 0, 0>("mfma 32x32x64 bf6 x16");
    run<0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 32>("mfma 16x16x128 bf6 x32");
    run<11, 0, 5, 0, 0, 0, 0, 0, 0, 0, 0, 0>("mfma 32x32 f16:bf6 2:1");
    run<0, 22, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 10>("mfma 16x16 f16:bf6 2:1");
    // the body kernel's ratios per 16 MFMA (per block: 256 f16 + 128 bf6 MFMA, 547 ds_read, 113 LDS-DMA per wave, 763 VALU of
    // which 16 cvt_pk32, 128 cvt_pk, 256 fma_mix, ~190 accvgpr, rest plain)
    run<16, 0, 0, 22, 0, 0, 0, 0, 0, 0, 0, 0>("mfma16 + ds_read 1.4/MFMA");
    run<16, 0, 0, 0, 0, 5, 0, 0, 0, 0, 0, 0, 0, 1>("mfma16 + LDS-DMA 0.3/MFMA (L2)");
    run<16, 0, 0, 0, 0, 0, 0, 1, 5, 5, 8, 4>("mfma16 + VALU 2/MFMA");
    run<11, 0, 5, 22, 0, 5, 0, 1, 5, 5, 8, 4, 0, 1>("body replica (32x32 shapes)");
    run<11, 0, 5, 22, 0, 0, 0, 1, 5, 5, 8, 4, 0, 1>("body replica, no DMA");
    run<11, 0, 5, 0, 0, 5, 0, 1, 5, 5, 8, 4, 0, 1>("body replica, no ds_read");
    run<11, 0, 5, 22, 0, 5, 0, 0, 0, 0, 0, 0, 0, 1>("body replica, no VALU");
    run<0, 22, 0, 22, 0, 5, 0, 1, 5, 5, 8, 4, 10, 1>("body replica in 16x16 shapes");
    run<0, 22, 5, 22, 0, 5, 0, 1, 5, 5, 8, 4, 0, 1>("body replica, f16 16x16 + bf6 32x32");
    // weights from global into registers, activations through LDS: half the ds_read, + ds_write, no LDS-DMA
    run<11, 0, 5, 11, 3, 0, 5, 1, 5, 5, 8, 4, 0, 1>("B-from-LDS variant (32x32)");
    run<0, 22, 0, 11, 3, 0, 5, 1, 5, 5, 8, 4, 10, 1>("B-from-LDS variant (16x16)");
    // the teacher chain: 16x16 shapes, per tile 3,732 MFMA (2/3 f16), 2,508 ds_read, 529 LDS-DMA per wave, 3,452 VALU
    run<0, 22, 0, 20, 0, 4, 0, 1, 8, 8, 10, 3, 10, 1>("chain replica (16x16)");
    run<0, 22, 0, 8, 3, 0, 4, 1, 8, 8, 10, 3, 10, 1>("chain, weights in registers");
    rsmi_shut_down();
    return 0;
}
