// Which form of a packed-fp32 VALU op goes wrong beside another process's nerf_chain_kernel (profiles/r04_gpu_sharing.txt)?
// Three kernels compute out[i] = (x[i] * c0 + y[i] * c1, x[i] * c2 + y[i] * c3) for 160,000 threads, 1,000 launches each:
//   S  v_pk_mul_f32 / v_pk_add_f32 with the constants as SGPR-PAIR operands (what hipcc's SLP vectorizer made of get_rays)
//   V  the same instructions with the constants copied into VGPR pairs first
//   F  scalar v_mul_f32 / v_add_f32 (the -fno-slp-vectorize form)
// and the host compares every value with the CPU's (exact: one rounding per operation, no contraction).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/pk_sgpr_probe.hip -o tools/pk_sgpr_probe
//   (python tools/gpu_sharing_check.py heavy-only in another process; see tools/pk_sgpr_run.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(const float* __restrict__ x, const float* __restrict__ y, float c0, float c1, float c2, float c3, int n,
                  float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float xv = x[i], yv = y[i];
    f32x2 r;
    if (MODE == 3) {
        // the shape of hipcc's get_rays code: ONE SGPR pair, rewritten by s_mov_b32 right behind the packed op that read it
        f32x2 a, b;
        f32x2 xx = {xv, xv}, yy = {yv, yv};
        asm volatile("s_mov_b32 s40, %4\n s_mov_b32 s41, %6\n v_pk_mul_f32 %0, %2, s[40:41]\n"
                     "s_mov_b32 s40, %5\n s_mov_b32 s41, %7\n v_pk_mul_f32 %1, %3, s[40:41]\n"
                     "s_mov_b32 s40, 0\n s_mov_b32 s41, 0\n s_nop 0\n v_pk_add_f32 %0, %0, %1"
                     : "=&v"(a), "=&v"(b)
                     : "v"(xx), "v"(yy), "s"(c0), "s"(c1), "s"(c2), "s"(c3)
                     : "s40", "s41");
        r = a;
    } else if (MODE == 2) {
        r[0] = __fadd_rn(__fmul_rn(xv, c0), __fmul_rn(yv, c1));
        r[1] = __fadd_rn(__fmul_rn(xv, c2), __fmul_rn(yv, c3));
    } else {
        f32x2 xx = {xv, xv}, yy = {yv, yv}, a, b;
        if (MODE == 0) {
            f32x2 ca = {c0, c2}, cb = {c1, c3};     // uniform values: "s" constraints put them into SGPR pairs
            asm volatile("v_pk_mul_f32 %0, %2, %4\n v_pk_mul_f32 %1, %3, %5\n s_nop 0\n v_pk_add_f32 %0, %0, %1"
                         : "=&v"(a), "=&v"(b)
                         : "v"(xx), "v"(yy), "s"(ca), "s"(cb));
        } else {
            f32x2 ca = {c0, c2}, cb = {c1, c3};
            asm volatile("v_pk_mul_f32 %0, %2, %4\n v_pk_mul_f32 %1, %3, %5\n s_nop 0\n v_pk_add_f32 %0, %0, %1"
                         : "=&v"(a), "=&v"(b)
                         : "v"(xx), "v"(yy), "v"(ca), "v"(cb));
        }
        r = a;
    }
    out[2 * i] = r[0];
    out[2 * i + 1] = r[1];
}

int main(int argc, char** argv) {
    const int n = 160000, reps = argc > 1 ? atoi(argv[1]) : 1000;
    std::vector<float> x(n), y(n), ref(2 * n), got(2 * n);
    srand(3);
    for (int i = 0; i < n; ++i) {
        x[i] = (float)rand() / RAND_MAX * 2 - 1;
        y[i] = (float)rand() / RAND_MAX * 2 - 1;
    }
    const float c[4] = {0.8412347f, -0.5406781f, 0.3128846f, 0.9497912f};
    for (int i = 0; i < n; ++i) {
        volatile float a = x[i] * c[0], b = y[i] * c[1], d = x[i] * c[2], e = y[i] * c[3];
        volatile float s0 = a + b, s1 = d + e;
        ref[2 * i] = s0;
        ref[2 * i + 1] = s1;
    }
    float *dx, *dy, *dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4); hipMalloc(&dout, 2 * n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dy, y.data(), n * 4, hipMemcpyHostToDevice);
    const char* names[4] = {"S  v_pk_* with SGPR-pair constants", "V  v_pk_* with VGPR-pair constants", "F  scalar v_mul / v_add",
                            "W  v_pk_* on ONE SGPR pair rewritten by s_mov right behind each use"};
    for (int mode = 0; mode < 4; ++mode) {
        long long bad = 0, bad_lo = 0, launches_bad = 0;
        for (int r = 0; r < reps; ++r) {
            hipMemsetAsync(dout, 0, 2 * n * 4, 0);
            if (mode == 0) k<0><<<(n + 255) / 256, 256>>>(dx, dy, c[0], c[1], c[2], c[3], n, dout);
            else if (mode == 1) k<1><<<(n + 255) / 256, 256>>>(dx, dy, c[0], c[1], c[2], c[3], n, dout);
            else if (mode == 2) k<2><<<(n + 255) / 256, 256>>>(dx, dy, c[0], c[1], c[2], c[3], n, dout);
            else k<3><<<(n + 255) / 256, 256>>>(dx, dy, c[0], c[1], c[2], c[3], n, dout);
            hipMemcpy(got.data(), dout, 2 * n * 4, hipMemcpyDeviceToHost);
            long long b = 0;
            for (int i = 0; i < 2 * n; ++i)
                if (got[i] != ref[i]) {
                    ++b;
                    if ((i & 1) == 0) ++bad_lo;
                }
            bad += b;
            launches_bad += b != 0;
        }
        printf("%-70s %d launches x %d threads: %lld wrong values (%lld in component 0) in %lld launches\n", names[mode], reps, n, bad, bad_lo,
               launches_bad);
        fflush(stdout);
    }
    return 0;
}
