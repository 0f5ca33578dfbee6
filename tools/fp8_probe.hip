// gfx950 probe of the e4m3 path of the body kernel's middle precision mode (FP16_FP8E):
// (1) v_cvt_scalef32_pk_fp8_f16: which half of the destination op_sel selects, the scale operand's sense, rounding,
//     saturation;  (2) v_mfma_scale_f32_32x32x64_f8f6f4 cbsz:0 blgp:0 (e4m3 x e4m3) with host-packed operands: byte e of lane
//     32h + r of A pairs with byte e of lane 32h + c of B, D layout, E8M0 scales, B operand in AGPRs;
// (3) issue time of 4 fp16 32x32x16 + 2 K=64 MFMAs per group: bf6 terms against e4m3 terms.
//   hipcc --offload-arch=gfx950 -O2 tools/fp8_probe.hip -o tools/fp8_probe && tools/fp8_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static float e4m3_val(int b) {
    const int s = (b >> 7) & 1, e = (b >> 3) & 15, m = b & 7;
    if (e == 15 && m == 7) return NAN;
    const float v = e == 0 ? ldexpf(m / 8.0f, -6) : ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -v : v;
}
// nearest e4m3 (ties to even), saturating at 448
static int e4m3_enc(float x) {
    const int s = signbit(x) ? 0x80 : 0;
    const float a = fabsf(x);
    int best = 0;
    float bd = INFINITY;
    for (int b = 0; b < 0x7f; ++b) {
        const float d = fabsf(e4m3_val(b) - a);
        if (d < bd || (d == bd && (b & 1) == 0)) { bd = d; best = b; }
    }
    return s | best;
}

__global__ void k_cvt(const uint32_t* src, uint32_t* out, float scale, int ovfl) {
    const int l = threadIdx.x;
    const uint32_t s0 = src[l * 2], s1 = src[l * 2 + 1];
    uint32_t d = 0xdeadbeef;
    if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1\n\ts_nop 2" ::: "memory");   // MODE.FP16_OVFL
    asm volatile(
        "v_cvt_scalef32_pk_fp8_f16 %0, %1, %3\n\t"
        "s_nop 1\n\t"
        "v_cvt_scalef32_pk_fp8_f16 %0, %2, %3 op_sel:[0,0,1]\n\t"
        "s_nop 1\n\t"
        : "+v"(d) : "v"(s0), "v"(s1), "v"(scale));
    out[l] = d;
}

__global__ void k_mfma8(const uint32_t* a, const uint32_t* b, float* d, uint32_t sa, uint32_t sb) {
    const int l = threadIdx.x;
    uint32_t av[8], bv[8];
    for (int i = 0; i < 8; ++i) { av[i] = a[l * 8 + i]; bv[i] = b[l * 8 + i]; }
    float o[16];
    asm volatile(
        "v_mov_b32 v40, %16\n\tv_mov_b32 v41, %17\n\tv_mov_b32 v42, %18\n\tv_mov_b32 v43, %19\n\t"
        "v_mov_b32 v44, %20\n\tv_mov_b32 v45, %21\n\tv_mov_b32 v46, %22\n\tv_mov_b32 v47, %23\n\t"
        "v_accvgpr_write_b32 a16, %24\n\tv_accvgpr_write_b32 a17, %25\n\tv_accvgpr_write_b32 a18, %26\n\tv_accvgpr_write_b32 a19, %27\n\t"
        "v_accvgpr_write_b32 a20, %28\n\tv_accvgpr_write_b32 a21, %29\n\tv_accvgpr_write_b32 a22, %30\n\tv_accvgpr_write_b32 a23, %31\n\t"
        "v_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\tv_mov_b32 v50, 0\n\tv_mov_b32 v51, 0\n\tv_mov_b32 v52, 0\n\tv_mov_b32 v53, 0\n\t"
        "v_mov_b32 v54, 0\n\tv_mov_b32 v55, 0\n\tv_mov_b32 v56, 0\n\tv_mov_b32 v57, 0\n\tv_mov_b32 v58, 0\n\tv_mov_b32 v59, 0\n\t"
        "v_mov_b32 v60, 0\n\tv_mov_b32 v61, 0\n\tv_mov_b32 v62, 0\n\tv_mov_b32 v63, 0\n\tv_mov_b32 v64, %32\n\tv_mov_b32 v65, %33\n\t"
        "s_nop 4\n\t"
        "v_mfma_scale_f32_32x32x64_f8f6f4 v[48:63], v[40:47], a[16:23], v[48:63], v64, v65 op_sel_hi:[0,0,0] cbsz:0 blgp:0\n\t"
        "s_nop 15\n\ts_nop 15\n\t"
        "v_mov_b32 %0, v48\n\tv_mov_b32 %1, v49\n\tv_mov_b32 %2, v50\n\tv_mov_b32 %3, v51\n\tv_mov_b32 %4, v52\n\tv_mov_b32 %5, v53\n\t"
        "v_mov_b32 %6, v54\n\tv_mov_b32 %7, v55\n\tv_mov_b32 %8, v56\n\tv_mov_b32 %9, v57\n\tv_mov_b32 %10, v58\n\tv_mov_b32 %11, v59\n\t"
        "v_mov_b32 %12, v60\n\tv_mov_b32 %13, v61\n\tv_mov_b32 %14, v62\n\tv_mov_b32 %15, v63\n\t"
        : "=v"(o[0]), "=v"(o[1]), "=v"(o[2]), "=v"(o[3]), "=v"(o[4]), "=v"(o[5]), "=v"(o[6]), "=v"(o[7]), "=v"(o[8]), "=v"(o[9]),
          "=v"(o[10]), "=v"(o[11]), "=v"(o[12]), "=v"(o[13]), "=v"(o[14]), "=v"(o[15])
        : "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]), "v"(av[4]), "v"(av[5]), "v"(av[6]), "v"(av[7]), "v"(bv[0]), "v"(bv[1]),
          "v"(bv[2]), "v"(bv[3]), "v"(bv[4]), "v"(bv[5]), "v"(bv[6]), "v"(bv[7]), "v"(sa), "v"(sb)
        : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57",
          "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23");
    for (int i = 0; i < 16; ++i) d[l * 16 + i] = o[i];
}

// issue time: per iteration 4 x (4 fp16 32x32x16 + 2 K=64 MFMAs of the given format) on one accumulator, one wave per SIMD
template <int FMT> __global__ __launch_bounds__(256, 1) void k_rate(float* out, int iters) {
    asm volatile(
        "v_mov_b32 v20, 0x3c003c00\n\tv_mov_b32 v21, 0x3c003c00\n\tv_mov_b32 v22, 0x3c003c00\n\tv_mov_b32 v23, 0x3c003c00\n\t"
        "v_mov_b32 v24, 0\n\tv_mov_b32 v25, 0\n\tv_mov_b32 v26, 0\n\tv_mov_b32 v27, 0\n\tv_mov_b32 v28, 0\n\tv_mov_b32 v29, 0\n\t"
        "v_mov_b32 v30, 0\n\tv_mov_b32 v31, 0\n\tv_mov_b32 v32, 0x7f7f7f7f\n\t"
        "s_mov_b32 s40, %0\n\t"
        "L_r_%=:\n\t"
        ".rept 4\n\t"
        "v_mfma_f32_32x32x16_f16 a[0:15], v[20:23], v[20:23], a[0:15]\n\t"
        "v_mfma_f32_32x32x16_f16 a[0:15], v[20:23], v[20:23], a[0:15]\n\t"
        "v_mfma_f32_32x32x16_f16 a[0:15], v[20:23], v[20:23], a[0:15]\n\t"
        "v_mfma_f32_32x32x16_f16 a[0:15], v[20:23], v[20:23], a[0:15]\n\t"
        ".if %1 == 0\n\t"
        "v_mfma_scale_f32_32x32x64_f8f6f4 a[0:15], v[24:31], v[24:31], a[0:15], v32, v32 op_sel_hi:[0,0,0]\n\t"
        "v_mfma_scale_f32_32x32x64_f8f6f4 a[0:15], v[24:31], v[24:31], a[0:15], v32, v32 op_sel_hi:[0,0,0]\n\t"
        ".else\n\t"
        "v_mfma_scale_f32_32x32x64_f8f6f4 a[0:15], v[24:29], v[24:29], a[0:15], v32, v32 op_sel_hi:[0,0,0] cbsz:3 blgp:3\n\t"
        "v_mfma_scale_f32_32x32x64_f8f6f4 a[0:15], v[24:29], v[24:29], a[0:15], v32, v32 op_sel_hi:[0,0,0] cbsz:3 blgp:3\n\t"
        ".endif\n\t"
        ".endr\n\t"
        "s_sub_u32 s40, s40, 1\n\ts_cmp_lg_u32 s40, 0\n\ts_cbranch_scc1 L_r_%=\n\t"
        "s_nop 15\n\t"
        :: "s"(iters), "i"(FMT)
        : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "v20", "v21", "v22", "v23",
          "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "s40", "scc");
    if (out) out[0] = 0;
}

int main() {
    // ---- (1) conversion: lane l converts (x0, x1) to the low half and (x2, x3) to the high half of one register ----
    static _Float16 vals[64][4];
    static uint32_t out[64];
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
            float v;
            if (l < 16) v = e4m3_val(8 * l + 2 * i + 1) * (i & 1 ? -1.f : 1.f);   // exactly representable values
            else if (l < 32) v = 1.0f + (4 * (l - 16) + i) / 64.0f;                 // rounding between 1 and 2 (step 1/8)
            else if (l < 48) v = (l - 31) * 40.0f + i;                              // up to 640: saturation
            else v = ldexpf(1.0f + i / 4.0f, -(l - 40));                            // small: subnormals below 2^-6
            vals[l][i] = (_Float16)v;
        }
    uint32_t *ds, *dout;
    hipMalloc((void**)&ds, sizeof vals); hipMalloc((void**)&dout, sizeof out);
    hipMemcpy(ds, vals, sizeof vals, hipMemcpyHostToDevice);
    const float scales[3] = {1.0f, 4.0f, 0.125f};
    for (int sc = 0; sc < 6; ++sc) {
        const int ovfl = sc >= 3;
        const float scale = scales[sc % 3];
        k_cvt<<<1, 64>>>(ds, dout, scale, ovfl);
        hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost);
        int bad_div = 0, bad_mul = 0;
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 4; ++i) {
                const int got = (out[l] >> (8 * i)) & 0xff;
                const float x = (float)vals[l][i];
                if (got != e4m3_enc(x / scale) && bad_div++ < 3)
                    printf("    lane %d el %d: %.9g / %g -> code 0x%02x (%g), host rule 0x%02x (%g)\n", l, i, x, scale, got, e4m3_val(got),
                           e4m3_enc(x / scale), e4m3_val(e4m3_enc(x / scale)));
                bad_mul += got != e4m3_enc(x * scale);
            }
        printf("cvt_scalef32_pk_fp8_f16 scale %g, MODE.FP16_OVFL %d: byte i = element i (low half first, op_sel[2] = high half), RNE, saturating: "
               "%d mismatches if dst = fp8(src / scale), %d if dst = fp8(src * scale)\n", scale, ovfl, bad_div, bad_mul);
        if (sc == 0) {
            printf("  lane 20 (rounding):");
            for (int i = 0; i < 4; ++i) printf(" %g->%g", (float)vals[20][i], e4m3_val((out[20] >> (8 * i)) & 0xff));
            printf("   lane 47 (saturation):");
            for (int i = 0; i < 4; ++i) printf(" %g->%g", (float)vals[47][i], e4m3_val((out[47] >> (8 * i)) & 0xff));
            printf("   lane 56 (subnormal):");
            for (int i = 0; i < 4; ++i) printf(" %g->%g", (float)vals[56][i], e4m3_val((out[56] >> (8 * i)) & 0xff));
            printf("\n");
        }
    }
    // ---- (2) MFMA with host-packed e4m3 ----
    {
        static int Ai[64][32], Bi[64][32]; static uint32_t A[64][8], B[64][8]; static float D[64][16];
        srand(3);
        for (int l = 0; l < 64; ++l) {
            memset(A[l], 0, 32); memset(B[l], 0, 32);
            for (int j = 0; j < 32; ++j) {
                Ai[l][j] = (rand() % 9) * (rand() & 1 ? -1 : 1); Bi[l][j] = (rand() % 9) * (rand() & 1 ? -1 : 1);
                A[l][j >> 2] |= (uint32_t)e4m3_enc((float)Ai[l][j]) << (8 * (j & 3));
                B[l][j >> 2] |= (uint32_t)e4m3_enc((float)Bi[l][j]) << (8 * (j & 3));
            }
        }
        uint32_t *da, *db; float* dd;
        hipMalloc((void**)&da, sizeof A); hipMalloc((void**)&db, sizeof B); hipMalloc((void**)&dd, sizeof D);
        hipMemcpy(da, A, sizeof A, hipMemcpyHostToDevice); hipMemcpy(db, B, sizeof B, hipMemcpyHostToDevice);
        for (int sc = 0; sc < 2; ++sc) {
            const int ea = sc ? -9 : 0, eb = sc ? 3 : 0;
            k_mfma8<<<1, 64>>>(da, db, dd, 0x01010101u * (127 + ea), 0x01010101u * (127 + eb));
            hipMemcpy(D, dd, sizeof D, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 16; ++r) {
                    const int row = 8 * (r / 4) + 4 * (l >> 5) + r % 4, col = l & 31;
                    long ref = 0;
                    for (int h = 0; h < 2; ++h) for (int j = 0; j < 32; ++j) ref += Ai[32 * h + row][j] * Bi[32 * h + col][j];
                    if (D[l][r] != (float)ldexp((double)ref, ea + eb)) { if (bad < 3) printf("  mfma8 mismatch lane %d r %d got %g want %g\n", l, r, D[l][r], ldexp((double)ref, ea + eb)); ++bad; }
                }
            printf("mfma 32x32x64 e4m3 x e4m3 (byte e of lane 32h+r pairs with byte e of lane 32h+c; B in AGPR), scales 2^%d 2^%d: %s (%d bad)\n",
                   ea, eb, bad ? "FAIL" : "ok", bad);
        }
    }
    // ---- (3) issue time ----
    for (int fmt = 0; fmt < 2; ++fmt) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 100000;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (fmt == 0) k_rate<0><<<256, 256>>>(nullptr, iters);
            if (fmt == 1) k_rate<1><<<256, 256>>>(nullptr, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const char* names[2] = {"4 fp16 + 2 e4m3 K=64", "4 fp16 + 2 bf6 K=64"};
        printf("rate %-22s: %.3f ms -> %.2f ns per group of 6 MFMAs (zeros: the clock is not the loaded one)\n", names[fmt], ms,
               ms * 1e6 / (iters * 4.0));
    }
    return 0;
}
