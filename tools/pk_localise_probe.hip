// Where does the wrong d.x of tools/get_rays_probe.hip come from?  The hand-written tail of that probe went wrong in every variant
// (also with no write-after-read behind the packed op), so this one stores the intermediates of the same sequence:
//   m0 = v_pk_mul_f32 (dy, dx) x s[40:41] = (dy c01, dx c10)        m1 = v_pk_mul_f32 (dy, dx) x s[40:41] = (dy c11, dx c00)
//   sm = v_pk_add_f32 m0, m1 op_sel:[0,1] op_sel_hi:[1,0]           fin = v_pk_add_f32 sm, s[40:41] neg_lo:[0,1] neg_hi:[0,1]
// in three forms of the constant operand: SGPR pair written by s_mov_b32 right in front (hipcc's form), SGPR pair written once at the
// top of the asm block (long before its use), VGPR pair.  All constants non-zero.  Every stored value is compared with the host's.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/pk_localise_probe.hip -o tools/pk_localise_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

template <int FORM>
__global__ void k(float c00, float c01, float c02, float c10, float c11, float c12, int W, float half_w, float half_h, float focal, int n,
                  float* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int jrow = i / W, icol = i - jrow * W;
    float dx = __fdiv_rn((float)icol - half_w, focal);
    float dy = -__fdiv_rn((float)jrow - half_h, focal);
    float r[8];
    if (FORM == 0)          // hipcc's form: one SGPR pair, rewritten in front of every use
        asm volatile(
            "v_mov_b32 v20, %[dy]\n v_mov_b32 v21, %[dx]\n"
            "s_mov_b32 s40, %[c01]\n s_mov_b32 s41, %[c10]\n"
            "v_pk_mul_f32 v[24:25], v[20:21], s[40:41]\n"
            "s_mov_b32 s40, %[c11]\n s_mov_b32 s41, %[c00]\n"
            "v_pk_mul_f32 v[26:27], v[20:21], s[40:41]\n"
            "v_pk_add_f32 v[28:29], v[24:25], v[26:27] op_sel:[0,1] op_sel_hi:[1,0]\n"
            "s_mov_b32 s40, %[c02]\n s_mov_b32 s41, %[c12]\n"
            "v_pk_add_f32 v[30:31], v[28:29], s[40:41] neg_lo:[0,1] neg_hi:[0,1]\n"
            "v_mov_b32 %[r0], v24\n v_mov_b32 %[r1], v25\n v_mov_b32 %[r2], v26\n v_mov_b32 %[r3], v27\n"
            "v_mov_b32 %[r4], v28\n v_mov_b32 %[r5], v29\n v_mov_b32 %[r6], v30\n v_mov_b32 %[r7], v31\n"
            : [r0] "=&v"(r[0]), [r1] "=&v"(r[1]), [r2] "=&v"(r[2]), [r3] "=&v"(r[3]), [r4] "=&v"(r[4]), [r5] "=&v"(r[5]), [r6] "=&v"(r[6]),
              [r7] "=&v"(r[7])
            : [dx] "v"(dx), [dy] "v"(dy), [c00] "s"(c00), [c01] "s"(c01), [c02] "s"(c02), [c10] "s"(c10), [c11] "s"(c11), [c12] "s"(c12)
            : "v20", "v21", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "s40", "s41");
    else if (FORM == 1)     // three SGPR pairs written once, 16 wait states before the first use
        asm volatile(
            "v_mov_b32 v20, %[dy]\n v_mov_b32 v21, %[dx]\n"
            "s_mov_b32 s40, %[c01]\n s_mov_b32 s41, %[c10]\n s_mov_b32 s42, %[c11]\n s_mov_b32 s43, %[c00]\n"
            "s_mov_b32 s44, %[c02]\n s_mov_b32 s45, %[c12]\n s_nop 15\n"
            "v_pk_mul_f32 v[24:25], v[20:21], s[40:41]\n"
            "v_pk_mul_f32 v[26:27], v[20:21], s[42:43]\n"
            "v_pk_add_f32 v[28:29], v[24:25], v[26:27] op_sel:[0,1] op_sel_hi:[1,0]\n"
            "v_pk_add_f32 v[30:31], v[28:29], s[44:45] neg_lo:[0,1] neg_hi:[0,1]\n"
            "v_mov_b32 %[r0], v24\n v_mov_b32 %[r1], v25\n v_mov_b32 %[r2], v26\n v_mov_b32 %[r3], v27\n"
            "v_mov_b32 %[r4], v28\n v_mov_b32 %[r5], v29\n v_mov_b32 %[r6], v30\n v_mov_b32 %[r7], v31\n"
            : [r0] "=&v"(r[0]), [r1] "=&v"(r[1]), [r2] "=&v"(r[2]), [r3] "=&v"(r[3]), [r4] "=&v"(r[4]), [r5] "=&v"(r[5]), [r6] "=&v"(r[6]),
              [r7] "=&v"(r[7])
            : [dx] "v"(dx), [dy] "v"(dy), [c00] "s"(c00), [c01] "s"(c01), [c02] "s"(c02), [c10] "s"(c10), [c11] "s"(c11), [c12] "s"(c12)
            : "v20", "v21", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "s40", "s41", "s42", "s43", "s44", "s45");
    else                    // the constants in VGPR pairs
        asm volatile(
            "v_mov_b32 v20, %[dy]\n v_mov_b32 v21, %[dx]\n"
            "v_mov_b32 v40, %[c01]\n v_mov_b32 v41, %[c10]\n v_mov_b32 v42, %[c11]\n v_mov_b32 v43, %[c00]\n"
            "v_mov_b32 v44, %[c02]\n v_mov_b32 v45, %[c12]\n"
            "v_pk_mul_f32 v[24:25], v[20:21], v[40:41]\n"
            "v_pk_mul_f32 v[26:27], v[20:21], v[42:43]\n"
            "v_pk_add_f32 v[28:29], v[24:25], v[26:27] op_sel:[0,1] op_sel_hi:[1,0]\n"
            "v_pk_add_f32 v[30:31], v[28:29], v[44:45] neg_lo:[0,1] neg_hi:[0,1]\n"
            "v_mov_b32 %[r0], v24\n v_mov_b32 %[r1], v25\n v_mov_b32 %[r2], v26\n v_mov_b32 %[r3], v27\n"
            "v_mov_b32 %[r4], v28\n v_mov_b32 %[r5], v29\n v_mov_b32 %[r6], v30\n v_mov_b32 %[r7], v31\n"
            : [r0] "=&v"(r[0]), [r1] "=&v"(r[1]), [r2] "=&v"(r[2]), [r3] "=&v"(r[3]), [r4] "=&v"(r[4]), [r5] "=&v"(r[5]), [r6] "=&v"(r[6]),
              [r7] "=&v"(r[7])
            : [dx] "v"(dx), [dy] "v"(dy), [c00] "s"(c00), [c01] "s"(c01), [c02] "s"(c02), [c10] "s"(c10), [c11] "s"(c11), [c12] "s"(c12)
            : "v20", "v21", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v40", "v41", "v42", "v43", "v44", "v45");
#pragma unroll
    for (int q = 0; q < 8; ++q) out[(size_t)q * n + i] = r[q];
}

int main(int argc, char** argv) {
    const int W = 400, n = W * W, reps = argc > 1 ? atoi(argv[1]) : 500;
    const float c00 = -0.6427876f, c01 = -0.3830222f, c02 = 0.6634139f, c10 = 0.7660444f, c11 = -0.3213938f, c12 = 0.5566704f;
    const float focal = 555.5555f * 1.37f, hw = 200.f, hh = 200.f;
    std::vector<float> ref(8 * (size_t)n), got(8 * (size_t)n), vdx(n), vdy(n);
    for (int i = 0; i < n; ++i) {
        const int j = i / W, ic = i - j * W;
        volatile float dx = ((float)ic - hw) / focal, dyp = ((float)j - hh) / focal;
        const float dy = -dyp;
        vdx[i] = dx; vdy[i] = dy;
        volatile float m0l = dy * c01, m0h = dx * c10, m1l = dy * c11, m1h = dx * c00;
        volatile float sl = m0l + m1h, sh = m0h + m1l;
        volatile float fl = sl - c02, fh = sh - c12;
        const float v[8] = {m0l, m0h, m1l, m1h, sl, sh, fl, fh};
        for (int q = 0; q < 8; ++q) ref[(size_t)q * n + i] = v[q];
    }
    float* out;
    hipMalloc(&out, 8 * (size_t)n * 4);
    const char* fn[3] = {"SGPR pair rewritten in front of each use (hipcc's form)", "SGPR pairs written once, s_nop 15 before use",
                         "VGPR pairs"};
    const char* qn[8] = {"m0.lo = dy c01", "m0.hi = dx c10", "m1.lo = dy c11", "m1.hi = dx c00", "sm.lo = m0.lo + m1.hi", "sm.hi = m0.hi + m1.lo",
                         "fin.lo = sm.lo - c02", "fin.hi = sm.hi - c12"};
    for (int f = 0; f < 3; ++f) {
        long long bad[8] = {0}, lb = 0;
        int shown = 0;
        for (int r = 0; r < reps; ++r) {
            hipMemsetAsync(out, 0xff, 8 * (size_t)n * 4, 0);
            if (f == 0) k<0><<<(n + 255) / 256, 256>>>(c00, c01, c02, c10, c11, c12, W, hw, hh, focal, n, out);
            else if (f == 1) k<1><<<(n + 255) / 256, 256>>>(c00, c01, c02, c10, c11, c12, W, hw, hh, focal, n, out);
            else k<2><<<(n + 255) / 256, 256>>>(c00, c01, c02, c10, c11, c12, W, hw, hh, focal, n, out);
            hipMemcpy(got.data(), out, 8 * (size_t)n * 4, hipMemcpyDeviceToHost);
            bool any = false;
            for (int q = 0; q < 8; ++q)
                for (int i = 0; i < n; ++i)
                    if (memcmp(&got[(size_t)q * n + i], &ref[(size_t)q * n + i], 4)) {
                        ++bad[q];
                        any = true;
                        if (shown < 6) {
                            ++shown;
                            printf("    launch %d thread %d (lane %d) %s: got %.9g expected %.9g   [dx %.9g dy %.9g; got/dx %.9g got/dy %.9g]\n", r, i,
                                   i & 63, qn[q], got[(size_t)q * n + i], ref[(size_t)q * n + i], vdx[i], vdy[i],
                                   got[(size_t)q * n + i] / vdx[i], got[(size_t)q * n + i] / vdy[i]);
                        }
                    }
            lb += any;
        }
        printf("%s: %d launches x %d threads, %lld launches with a wrong value\n", fn[f], reps, n, lb);
        for (int q = 0; q < 8; ++q) printf("    %-24s %lld wrong\n", qn[q], bad[q]);
        fflush(stdout);
    }
    return 0;
}
