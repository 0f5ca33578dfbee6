// The library's nerf_get_rays_kernel (csrc/nerf_kernels.hip, copied verbatim) in a STANDALONE process, compiled WITH the SLP
// vectorizer (the packed-fp32 form that went wrong inside the PyTorch process, profiles/r04_gpu_sharing.txt): does the instruction
// sequence alone reproduce the wrong d.x beside another process's nerf_chain_kernel, or does it take the library's process?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/get_rays_probe.hip -o tools/get_rays_probe      (no -fno-slp-vectorize)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void nerf_get_rays_kernel(float c00, float c01, float c02, float c03, float c10, float c11, float c12,
                                     float c13, float c20, float c21, float c22, float c23, int W, float half_w,
                                     float half_h, float focal, int pix_begin, int n, float* __restrict__ rays_o,
                                     float* __restrict__ rays_d) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int pix = pix_begin + i;
    int jrow = pix / W, icol = pix - jrow * W;
    float dx = __fdiv_rn((float)icol - half_w, focal);
    float dy = -__fdiv_rn((float)jrow - half_h, focal);
    const float c[12] = {c00, c01, c02, c03, c10, c11, c12, c13, c20, c21, c22, c23};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float s = __fadd_rn(__fmul_rn(dx, c[4 * k + 0]), __fmul_rn(dy, c[4 * k + 1]));
        rays_d[(size_t)i * 3 + k] = __fadd_rn(s, __fmul_rn(-1.0f, c[4 * k + 2]));
        rays_o[(size_t)i * 3 + k] = c[4 * k + 3];
    }
}

// The tail of that kernel by hand (fixed registers), to test ONE hypothesis: the VALU op right behind a packed-fp32 op overwrites one of
// the packed op's source VGPRs (here: v_pk_mul_f32 v[26:27], v[20:21], s[40:41] followed by v_mul_f32 v21, ...), and the packed op's last
// lanes read the NEW value.  NOPS > 0 puts s_nop NOPS-1 between the two; SWAP = 1 lets the overwrite go to a copy (no WAR at all).
template <int NOPS, int SWAP>
__global__ void tail_kernel(float c00, float c01, float c02, float c10, float c11, float c12, float c20, float c21, float c22, int W,
                            float half_w, float half_h, float focal, int n, float* __restrict__ rays_d) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int jrow = i / W, icol = i - jrow * W;
    float dx = __fdiv_rn((float)icol - half_w, focal);
    float dy = -__fdiv_rn((float)jrow - half_h, focal);
    float o0, o1, o2;
    asm volatile(
        "v_mov_b32 v20, %[dy]\n v_mov_b32 v21, %[dx]\n"
        "s_mov_b32 s40, %[c01]\n s_mov_b32 s41, %[c10]\n"
        "v_pk_mul_f32 v[24:25], v[20:21], s[40:41]\n"            // (dy c01, dx c10)
        "s_mov_b32 s40, %[c11]\n s_mov_b32 s41, %[c00]\n"
        "v_pk_mul_f32 v[26:27], v[20:21], s[40:41]\n"            // (dy c11, dx c00)
        ".if %c[nops] > 0\n s_nop (%c[nops] - 1)\n .endif\n"
        ".if %c[swap]\n v_mul_f32 v23, %[c20], v21\n v_mul_f32 v22, %[c21], v20\n"
        ".else\n v_mul_f32 v21, %[c20], v21\n v_mul_f32 v20, %[c21], v20\n .endif\n"     // the compiler's form: overwrites v21, v20
        "v_pk_add_f32 v[24:25], v[24:25], v[26:27] op_sel:[0,1] op_sel_hi:[1,0]\n"   // (dy c01 + dx c00, dx c10 + dy c11)
        "s_mov_b32 s40, %[c02]\n s_mov_b32 s41, %[c12]\n"
        ".if %c[swap]\n v_add_f32 v22, v23, v22\n .else\n v_add_f32 v20, v21, v20\n .endif\n"
        "v_pk_add_f32 v[24:25], v[24:25], s[40:41] neg_lo:[0,1] neg_hi:[0,1]\n"
        ".if %c[swap]\n v_subrev_f32 v22, %[c22], v22\n v_mov_b32 %[o2], v22\n .else\n v_subrev_f32 v20, %[c22], v20\n v_mov_b32 %[o2], v20\n .endif\n"
        "v_mov_b32 %[o0], v24\n v_mov_b32 %[o1], v25\n"
        : [o0] "=&v"(o0), [o1] "=&v"(o1), [o2] "=&v"(o2)
        : [dx] "v"(dx), [dy] "v"(dy), [c00] "s"(c00), [c01] "s"(c01), [c02] "s"(c02), [c10] "s"(c10), [c11] "s"(c11), [c12] "s"(c12),
          [c20] "s"(c20), [c21] "s"(c21), [c22] "s"(c22), [nops] "i"(NOPS), [swap] "i"(SWAP)
        : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "s40", "s41");
    rays_d[(size_t)i * 3 + 0] = o0;
    rays_d[(size_t)i * 3 + 1] = o1;
    rays_d[(size_t)i * 3 + 2] = o2;
}

int main(int argc, char** argv) {
    const int W = 400, n = W * W, reps = argc > 1 ? atoi(argv[1]) : 2000;
    const float c[12] = {-0.6427876f, -0.3830222f, 0.6634139f, 2.6536556f, 0.7660444f, -0.3213938f, 0.5566704f, 2.2266816f,
                         0.0f, 0.8660254f, 0.5f, 2.0f};
    const float focal = 555.5555f * 1.37f, hw = 200.f, hh = 200.f;
    std::vector<float> ref(3 * (size_t)n), got(3 * (size_t)n);
    for (int i = 0; i < n; ++i) {
        const int j = i / W, ic = i - j * W;
        volatile float dx = ((float)ic - hw) / focal, dyp = ((float)j - hh) / focal;
        const float dy = -dyp;
        for (int k = 0; k < 3; ++k) {
            volatile float a = dx * c[4 * k], b = dy * c[4 * k + 1], s = a + b, m = -1.0f * c[4 * k + 2];
            volatile float r = s + m;
            ref[(size_t)i * 3 + k] = r;
        }
    }
    float *ro, *rd;
    hipMalloc(&ro, 3 * n * 4);
    hipMalloc(&rd, 3 * n * 4);
    long long bad = 0, bad0 = 0, lb = 0;
    int shown = 0;
    for (int r = 0; r < reps; ++r) {
        hipMemsetAsync(rd, 0, 3 * n * 4, 0);
        nerf_get_rays_kernel<<<(n + 255) / 256, 256>>>(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], c[9], c[10], c[11], W, hw, hh, focal,
                                                      0, n, ro, rd);
        hipMemcpy(got.data(), rd, 3 * (size_t)n * 4, hipMemcpyDeviceToHost);
        long long b = 0;
        for (size_t i = 0; i < 3 * (size_t)n; ++i)
            if (got[i] != ref[i]) {
                ++b;
                if (i % 3 == 0) ++bad0;
                if (shown < 12 && i % 3 == 0) {     // which operand was wrong?  d.x = fl(fl(dx A) + fl(dy B)) + fl(-1 C) over all constants
                    ++shown;
                    const int ray = (int)(i / 3), j = ray / W, ic = ray - j * W;
                    volatile float dx = ((float)ic - hw) / focal, dyp = ((float)j - hh) / focal;
                    const float dy = -dyp;
                    printf("  launch %d ray %d (lane %d of its wave): got %.9g, expected %.9g;", r, ray, ray & 63, got[i], ref[i]);
                    int hits = 0;
                    for (int A = 0; A < 12; ++A)
                        for (int Bq = 0; Bq < 12; ++Bq)
                            for (int Cq = 0; Cq < 12; ++Cq) {
                                volatile float a = dx * c[A], bb = dy * c[Bq], ss = a + bb, m = -1.0f * c[Cq];
                                volatile float v = ss + m;
                                if (v == got[i] && hits < 4) {
                                    ++hits;
                                    printf(" = dx c[%d] + dy c[%d] - c[%d]", A, Bq, Cq);
                                }
                            }
                    printf("%s\n", hits ? "   (expected: dx c[0] + dy c[1] - c[2])" : "   (no combination of the constants)");
                }
            }
        bad += b;
        lb += b != 0;
    }
    printf("nerf_get_rays_kernel (SLP build) in a standalone process: %d launches x %d rays: %lld wrong values (%lld of them d.x) in %lld launches\n",
           reps, n, bad, bad0, lb);
    const char* vn[4] = {"hand-written tail, v_mul overwrites the packed op's source right behind it (the compiler's order)",
                         "... with s_nop 0 between them", "... with s_nop 3 between them", "... the v_mul writes another register (no WAR)"};
    for (int v = 0; v < 4; ++v) {
        long long vb = 0, vb0 = 0, vl = 0;
        for (int r = 0; r < reps; ++r) {
            hipMemsetAsync(rd, 0, 3 * n * 4, 0);
#define TK(N, S) tail_kernel<N, S><<<(n + 255) / 256, 256>>>(c[0], c[1], c[2], c[4], c[5], c[6], c[8], c[9], c[10], W, hw, hh, focal, n, rd)
            if (v == 0) TK(0, 0);
            else if (v == 1) TK(1, 0);
            else if (v == 2) TK(4, 0);
            else TK(0, 1);
            hipMemcpy(got.data(), rd, 3 * (size_t)n * 4, hipMemcpyDeviceToHost);
            long long b = 0;
            for (size_t i = 0; i < 3 * (size_t)n; ++i)
                if (got[i] != ref[i]) {
                    ++b;
                    if (i % 3 == 0) ++vb0;
                }
            vb += b;
            vl += b != 0;
        }
        printf("%-100s %d launches: %lld wrong values (%lld of them d.x) in %lld launches\n", vn[v], reps, vb, vb0, vl);
        fflush(stdout);
    }
    return 0;
}
