// gfx950 probe of the bf6 (e3m2) path: (1) v_cvt_scalef32_pk32_bf6_f16: element order, bit packing, scale operand and
// rounding; (2) v_mfma_scale_f32_16x16x128_f8f6f4 cbsz:3 blgp:3 with host-packed operands (little-endian 6-bit fields);
// (3) issue time of the fp16 + bf6 mix against the fp16 + fp8 mix.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

__global__ void k_cvt(const uint32_t* src, uint32_t* out, float scale) {
    const int l = threadIdx.x;
    uint32_t s[16];
    for (int i = 0; i < 16; ++i) s[i] = src[l * 16 + i];
    uint32_t o0, o1, o2, o3, o4, o5;
    asm volatile(
        "v_mov_b32 v40, %6\n\tv_mov_b32 v41, %7\n\tv_mov_b32 v42, %8\n\tv_mov_b32 v43, %9\n\t"
        "v_mov_b32 v44, %10\n\tv_mov_b32 v45, %11\n\tv_mov_b32 v46, %12\n\tv_mov_b32 v47, %13\n\t"
        "v_mov_b32 v48, %14\n\tv_mov_b32 v49, %15\n\tv_mov_b32 v50, %16\n\tv_mov_b32 v51, %17\n\t"
        "v_mov_b32 v52, %18\n\tv_mov_b32 v53, %19\n\tv_mov_b32 v54, %20\n\tv_mov_b32 v55, %21\n\t"
        "v_mov_b32 v56, %22\n\t"
        "s_nop 1\n\t"
        "v_cvt_scalef32_pk32_bf6_f16 v[60:65], v[40:55], v56\n\t"
        "s_nop 4\n\t"
        "v_mov_b32 %0, v60\n\tv_mov_b32 %1, v61\n\tv_mov_b32 %2, v62\n\tv_mov_b32 %3, v63\n\tv_mov_b32 %4, v64\n\tv_mov_b32 %5, v65\n\t"
        : "=v"(o0), "=v"(o1), "=v"(o2), "=v"(o3), "=v"(o4), "=v"(o5)
        : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]), "v"(s[5]), "v"(s[6]), "v"(s[7]), "v"(s[8]), "v"(s[9]), "v"(s[10]),
          "v"(s[11]), "v"(s[12]), "v"(s[13]), "v"(s[14]), "v"(s[15]), "v"(scale)
        : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56",
          "v60", "v61", "v62", "v63", "v64", "v65");
    out[l * 6 + 0] = o0; out[l * 6 + 1] = o1; out[l * 6 + 2] = o2; out[l * 6 + 3] = o3; out[l * 6 + 4] = o4; out[l * 6 + 5] = o5;
}

__global__ void k_mfma6(const uint32_t* a, const uint32_t* b, float* d, uint32_t sa, uint32_t sb) {
    const int l = threadIdx.x;
    uint32_t av[6], bv[6];
    for (int i = 0; i < 6; ++i) { av[i] = a[l * 6 + i]; bv[i] = b[l * 6 + i]; }
    float o0, o1, o2, o3;
    asm volatile(
        "v_mov_b32 v40, %4\n\tv_mov_b32 v41, %5\n\tv_mov_b32 v42, %6\n\tv_mov_b32 v43, %7\n\tv_mov_b32 v44, %8\n\tv_mov_b32 v45, %9\n\t"
        "v_accvgpr_write_b32 a16, %10\n\tv_accvgpr_write_b32 a17, %11\n\tv_accvgpr_write_b32 a18, %12\n\t"
        "v_accvgpr_write_b32 a19, %13\n\tv_accvgpr_write_b32 a20, %14\n\tv_accvgpr_write_b32 a21, %15\n\t"
        "v_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\tv_mov_b32 v50, 0\n\tv_mov_b32 v51, 0\n\tv_mov_b32 v52, %16\n\tv_mov_b32 v53, %17\n\t"
        "s_nop 4\n\t"
        "v_mfma_scale_f32_16x16x128_f8f6f4 v[48:51], v[40:45], a[16:21], v[48:51], v52, v53 op_sel_hi:[0,0,0] cbsz:3 blgp:3\n\t"
        "s_nop 15\n\t"
        "v_mov_b32 %0, v48\n\tv_mov_b32 %1, v49\n\tv_mov_b32 %2, v50\n\tv_mov_b32 %3, v51\n\t"
        : "=v"(o0), "=v"(o1), "=v"(o2), "=v"(o3)
        : "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]), "v"(av[4]), "v"(av[5]), "v"(bv[0]), "v"(bv[1]), "v"(bv[2]), "v"(bv[3]),
          "v"(bv[4]), "v"(bv[5]), "v"(sa), "v"(sb)
        : "v40", "v41", "v42", "v43", "v44", "v45", "v48", "v49", "v50", "v51", "v52", "v53", "a16", "a17", "a18", "a19", "a20", "a21");
    d[l * 4 + 0] = o0; d[l * 4 + 1] = o1; d[l * 4 + 2] = o2; d[l * 4 + 3] = o3;
}

// issue time: per iteration 8 x (2 fp16 MFMAs + 1 K=128 MFMA of the given format), one wave per SIMD
template <int FMT> __global__ __launch_bounds__(256, 1) void k_rate(float* out, int iters) {
    asm volatile(
        "v_mov_b32 v20, 0x3c003c00\n\tv_mov_b32 v21, 0x3c003c00\n\tv_mov_b32 v22, 0x3c003c00\n\tv_mov_b32 v23, 0x3c003c00\n\t"
        "v_mov_b32 v24, 0\n\tv_mov_b32 v25, 0\n\tv_mov_b32 v26, 0\n\tv_mov_b32 v27, 0\n\tv_mov_b32 v28, 0\n\tv_mov_b32 v29, 0\n\t"
        "v_mov_b32 v30, 0\n\tv_mov_b32 v31, 0\n\tv_mov_b32 v32, 0x7f7f7f7f\n\t"
        "v_mov_b32 v0, 0\n\tv_mov_b32 v1, 0\n\tv_mov_b32 v2, 0\n\tv_mov_b32 v3, 0\n\tv_mov_b32 v4, 0\n\tv_mov_b32 v5, 0\n\tv_mov_b32 v6, 0\n\tv_mov_b32 v7, 0\n\t"
        "s_mov_b32 s40, %0\n\t"
        "L_r_%=:\n\t"
        ".rept 8\n\t"
        "v_mfma_f32_16x16x32_f16 v[0:3], v[20:23], v[20:23], v[0:3]\n\t"
        "v_mfma_f32_16x16x32_f16 v[4:7], v[20:23], v[20:23], v[4:7]\n\t"
        ".if %1 == 0\n\t"
        "v_mfma_scale_f32_16x16x128_f8f6f4 v[0:3], v[24:31], v[24:31], v[0:3], v32, v32 op_sel_hi:[0,0,0] blgp:1\n\t"
        ".elseif %1 == 1\n\t"
        "v_mfma_scale_f32_16x16x128_f8f6f4 v[0:3], v[24:29], v[24:29], v[0:3], v32, v32 op_sel_hi:[0,0,0] cbsz:3 blgp:3\n\t"
        ".elseif %1 == 2\n\t"
        "v_mfma_f32_16x16x128_f8f6f4 v[0:3], v[24:29], v[24:29], v[0:3] cbsz:3 blgp:3\n\t"
        ".else\n\t"
        "v_mfma_f32_16x16x128_f8f6f4 v[0:3], v[24:31], v[24:31], v[0:3] blgp:1\n\t"
        ".endif\n\t"
        ".endr\n\t"
        "s_sub_u32 s40, s40, 1\n\ts_cmp_lg_u32 s40, 0\n\ts_cbranch_scc1 L_r_%=\n\t"
        "s_nop 15\n\t"
        :: "s"(iters), "i"(FMT)
        : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30",
          "v31", "v32", "s40", "scc");
    if (out) out[0] = 0;
}

static float e3m2_val(int b) {
    int s = (b >> 5) & 1, e = (b >> 2) & 7, m = b & 3;
    float v = e == 0 ? ldexpf(m / 4.0f, -2) : ldexpf(1.0f + m / 4.0f, e - 3);
    return s ? -v : v;
}
static int e3m2_enc(float x) {  // exact inputs only
    for (int b = 0; b < 64; ++b) if (e3m2_val(b) == x) return b;
    return -1;
}
static void pack6(const int* el, uint32_t* out6) {  // 32 6-bit fields, element i at bits [6i, 6i+6)
    memset(out6, 0, 24);
    for (int i = 0; i < 32; ++i) {
        const int bit = 6 * i;
        out6[bit >> 5] |= (uint32_t)el[i] << (bit & 31);
        if ((bit & 31) > 26) out6[(bit >> 5) + 1] |= (uint32_t)el[i] >> (32 - (bit & 31));
    }
}
static int get6(const uint32_t* w, int i) {
    const int bit = 6 * i;
    uint64_t v = w[bit >> 5] | ((uint64_t)((bit >> 5) + 1 < 6 ? w[(bit >> 5) + 1] : 0) << 32);
    return (int)((v >> (bit & 31)) & 63);
}

int main() {
    // ---- (1) conversion ----
    static uint32_t src[64][16], out[64][6];
    static _Float16 vals[64][32];
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 32; ++i) {
            // lane 0: 32 distinct exactly representable values; lane 1: rounding cases; others: scaled copies
            float v = l == 0 ? e3m2_val(1 + (i % 31)) * (i & 1 ? -1.f : 1.f) : l == 1 ? 1.0f + i / 32.0f : (i + 1) * 0.5f;
            vals[l][i] = (_Float16)v;
        }
    memcpy(src, vals, sizeof src);
    uint32_t *ds, *dout;
    hipMalloc((void**)&ds, sizeof src); hipMalloc((void**)&dout, sizeof out);
    hipMemcpy(ds, src, sizeof src, hipMemcpyHostToDevice);
    const float scales[3] = {1.0f, 2.0f, 0.25f};
    for (int sc = 0; sc < 3; ++sc) {
        k_cvt<<<1, 64>>>(ds, dout, scales[sc]);
        hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost);
        printf("cvt scale=%g lane0 (in -> out):", scales[sc]);
        for (int i = 0; i < 8; ++i) printf(" %g->%g", (float)vals[0][i], e3m2_val(get6(out[0], i)));
        printf(" ... el31 %g->%g\n", (float)vals[0][31], e3m2_val(get6(out[0], 31)));
        if (sc == 0) {
            int bad = 0;
            for (int i = 0; i < 32; ++i) bad += e3m2_val(get6(out[0], i)) != (float)vals[0][i];
            printf("  element order / little-endian 6-bit packing at scale 1: %s (%d mismatches)\n", bad ? "DIFFERENT" : "ok", bad);
            printf("  rounding lane1:");
            for (int i = 0; i < 32; i += 3) printf(" %g->%g", (float)vals[1][i], e3m2_val(get6(out[1], i)));
            printf("\n  lane2 (0.5 .. 16):");
            for (int i = 0; i < 32; i += 4) printf(" %g->%g", (float)vals[2][i], e3m2_val(get6(out[2], i)));
            printf("\n");
        }
    }
    // ---- (2) MFMA with host-packed bf6 ----
    {
        static int Ai[64][32], Bi[64][32]; static uint32_t A[64][6], B[64][6]; static float D[64][4];
        const float cand[9] = {0, 1, 2, 3, 4, 5, 6, 7, 8};
        srand(3);
        for (int l = 0; l < 64; ++l) {
            int ea[32], eb[32];
            for (int j = 0; j < 32; ++j) {
                Ai[l][j] = (rand() % 9) * (rand() & 1 ? -1 : 1); Bi[l][j] = (rand() % 9) * (rand() & 1 ? -1 : 1);
                ea[j] = e3m2_enc(cand[abs(Ai[l][j])]) | (Ai[l][j] < 0 ? 32 : 0);
                eb[j] = e3m2_enc(cand[abs(Bi[l][j])]) | (Bi[l][j] < 0 ? 32 : 0);
            }
            pack6(ea, A[l]); pack6(eb, B[l]);
        }
        uint32_t *da, *db; float* dd;
        hipMalloc((void**)&da, sizeof A); hipMalloc((void**)&db, sizeof B); hipMalloc((void**)&dd, sizeof D);
        hipMemcpy(da, A, sizeof A, hipMemcpyHostToDevice); hipMemcpy(db, B, sizeof B, hipMemcpyHostToDevice);
        for (int sc = 0; sc < 2; ++sc) {
            const int ea = sc ? -9 : 0, eb = sc ? 3 : 0;
            k_mfma6<<<1, 64>>>(da, db, dd, 0x01010101u * (127 + ea), 0x01010101u * (127 + eb));
            hipMemcpy(D, dd, sizeof D, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int l = 0; l < 64; ++l)
                for (int i = 0; i < 4; ++i) {
                    const int row = 4 * (l >> 4) + i, col = l & 15;
                    long ref = 0;
                    for (int q = 0; q < 4; ++q) for (int j = 0; j < 32; ++j) ref += Ai[q * 16 + row][j] * Bi[q * 16 + col][j];
                    if (D[l][i] != (float)ldexp((double)ref, ea + eb)) { if (bad < 3) printf("  mfma6 mismatch lane %d i %d got %g want %g\n", l, i, D[l][i], ldexp((double)ref, ea + eb)); ++bad; }
                }
            printf("mfma bf6 x bf6 (B in AGPR), scales 2^%d 2^%d: %s (%d bad)\n", ea, eb, bad ? "FAIL" : "ok", bad);
        }
    }
    // ---- (3) issue time ----
    for (int fmt = 0; fmt < 4; ++fmt) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 200000;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (fmt == 0) k_rate<0><<<256, 256>>>(nullptr, iters);
            if (fmt == 1) k_rate<1><<<256, 256>>>(nullptr, iters);
            if (fmt == 2) k_rate<2><<<256, 256>>>(nullptr, iters);
            if (fmt == 3) k_rate<3><<<256, 256>>>(nullptr, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const char* names[4] = {"2 fp16 + 1 fp8 scaled", "2 fp16 + 1 bf6 scaled", "2 fp16 + 1 bf6 unscaled", "2 fp16 + 1 fp8 unscaled"};
        printf("rate %-26s: %.3f ms -> %.2f ns per group (at 2.0 GHz: %.1f cycles)\n", names[fmt], ms, ms * 1e6 / (iters * 8.0), ms * 1e6 / (iters * 8.0) * 2.0);
    }
    return 0;
}
