// Checks on gfx950 the instruction forms the generated body loop (csrc/gen/body_gen.py) relies on, against a host
// evaluation: (1) v_mfma_scale_f32_16x16x128_f8f6f4 with the B operand in AGPRs, C/D in VGPRs and large E8M0 shifts
// (2^-24, 2^-12); (2) the epilogue chain v_cvt_pk_f16_f32 / v_fma_mix_f32 (t - hi) / v_cvt_pk_bf8_f32 / v_accvgpr_write;
// (3) v_mfma_f32_16x16x32_f16 with an AGPR B operand and C != D.
// hipcc --offload-arch=gfx950 -O2 tools/asm_semantics_test.hip -o /tmp/asm_sem && /tmp/asm_sem
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

__global__ void k_mfma8(const uint32_t* a, const uint32_t* b, const float* c, float* d, uint32_t sa, uint32_t sb) {
    const int l = threadIdx.x;
    uint32_t av[8], bv[8];
    float cv[4];
    for (int i = 0; i < 8; ++i) { av[i] = a[l * 8 + i]; bv[i] = b[l * 8 + i]; }
    for (int i = 0; i < 4; ++i) cv[i] = c[l * 4 + i];
    float o0, o1, o2, o3;
    asm volatile(
        "v_mov_b32 v[40], %4\n\tv_mov_b32 v[41], %5\n\tv_mov_b32 v[42], %6\n\tv_mov_b32 v[43], %7\n\t"
        "v_mov_b32 v[44], %8\n\tv_mov_b32 v[45], %9\n\tv_mov_b32 v[46], %10\n\tv_mov_b32 v[47], %11\n\t"
        "v_accvgpr_write_b32 a[16], %12\n\tv_accvgpr_write_b32 a[17], %13\n\tv_accvgpr_write_b32 a[18], %14\n\t"
        "v_accvgpr_write_b32 a[19], %15\n\tv_accvgpr_write_b32 a[20], %16\n\tv_accvgpr_write_b32 a[21], %17\n\t"
        "v_accvgpr_write_b32 a[22], %18\n\tv_accvgpr_write_b32 a[23], %19\n\t"
        "v_mov_b32 v[48], %20\n\tv_mov_b32 v[49], %21\n\tv_mov_b32 v[50], %22\n\tv_mov_b32 v[51], %23\n\t"
        "v_mov_b32 v[52], %24\n\tv_mov_b32 v[53], %25\n\t"
        "s_nop 4\n\t"
        "v_mfma_scale_f32_16x16x128_f8f6f4 v[48:51], v[40:47], a[16:23], v[48:51], v52, v53 op_sel_hi:[0,0,0] blgp:1\n\t"
        "s_nop 15\n\t"
        "v_mov_b32 %0, v[48]\n\tv_mov_b32 %1, v[49]\n\tv_mov_b32 %2, v[50]\n\tv_mov_b32 %3, v[51]\n\t"
        : "=v"(o0), "=v"(o1), "=v"(o2), "=v"(o3)
        : "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]), "v"(av[4]), "v"(av[5]), "v"(av[6]), "v"(av[7]), "v"(bv[0]), "v"(bv[1]),
          "v"(bv[2]), "v"(bv[3]), "v"(bv[4]), "v"(bv[5]), "v"(bv[6]), "v"(bv[7]), "v"(cv[0]), "v"(cv[1]), "v"(cv[2]), "v"(cv[3]),
          "v"(sa), "v"(sb)
        : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "a16", "a17", "a18",
          "a19", "a20", "a21", "a22", "a23");
    d[l * 4 + 0] = o0; d[l * 4 + 1] = o1; d[l * 4 + 2] = o2; d[l * 4 + 3] = o3;
}

// epilogue chain on 4 values per lane -> h01, h23, qa, qr (read back through AGPRs)
__global__ void k_epi(const float* t, uint32_t* out) {
    const int l = threadIdx.x;
    float t0 = t[l * 4], t1 = t[l * 4 + 1], t2 = t[l * 4 + 2], t3 = t[l * 4 + 3];
    uint32_t h01, h23, qa, qr;
    asm volatile(
        "s_mov_b32 s68, 0xbf800000\n\t"
        "v_cvt_pk_f16_f32 v60, %4, %5\n\t"
        "v_cvt_pk_f16_f32 v61, %6, %7\n\t"
        "v_accvgpr_write_b32 a30, v60\n\t"
        "v_accvgpr_write_b32 a31, v61\n\t"
        "v_cvt_pk_bf8_f32 v66, %4, %5\n\t"
        "v_cvt_pk_bf8_f32 v66, %6, %7 op_sel:[0,0,1]\n\t"
        "v_accvgpr_write_b32 a32, v66\n\t"
        "v_fma_mix_f32 v62, v60, s68, %4 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 v63, v60, s68, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 v64, v61, s68, %6 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 v65, v61, s68, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_cvt_pk_bf8_f32 v67, v62, v63\n\t"
        "v_cvt_pk_bf8_f32 v67, v64, v65 op_sel:[0,0,1]\n\t"
        "v_accvgpr_write_b32 a33, v67\n\t"
        "s_nop 2\n\t"
        "v_accvgpr_read_b32 %0, a30\n\tv_accvgpr_read_b32 %1, a31\n\tv_accvgpr_read_b32 %2, a32\n\tv_accvgpr_read_b32 %3, a33\n\t"
        : "=v"(h01), "=v"(h23), "=v"(qa), "=v"(qr)
        : "v"(t0), "v"(t1), "v"(t2), "v"(t3)
        : "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "a30", "a31", "a32", "a33", "s68");
    out[l * 4] = h01; out[l * 4 + 1] = h23; out[l * 4 + 2] = qa; out[l * 4 + 3] = qr;
}

static float e4m3_val(uint8_t b) {
    int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -v : v;
}
static float e5m2_val(uint8_t b) {
    int s = b >> 7, e = (b >> 2) & 31, m = b & 3;
    float v = e == 0 ? ldexpf((float)m, -16) : ldexpf(1.0f + m / 4.0f, e - 15);
    return s ? -v : v;
}
static uint8_t f32_to_e5m2_rne(float x) {  // exact: nearest e5m2 (ties to even mantissa), subnormals included, no overflow here
    const uint8_t s = signbit(x) ? 0x80 : 0;
    const double a = fabs((double)x);
    int best = 0; double bd = 1e300;
    for (int b = 0; b < 0x7c; ++b) {
        const double d = fabs((double)e5m2_val((uint8_t)b) - a);
        if (d < bd || (d == bd && !(b & 1))) { bd = d; best = b; }
    }
    return s | (uint8_t)best;
}

int main() {
    int fails = 0;
    // ---- (1) MFMA8 -------------------------------------------------------------------------------
    static uint8_t A[64][32], B[64][32];
    static float C[64][4], D[64][4];
    for (int trial = 0; trial < 4; ++trial) {
        srand(10 + trial);
        for (int l = 0; l < 64; ++l) {
            for (int j = 0; j < 32; ++j) {
                // e4m3 bytes with exponent field 1..14 (no NaN), e5m2 bytes with exponent 1..29
                A[l][j] = (uint8_t)(((rand() & 1) << 7) | ((1 + rand() % 14) << 3) | (rand() & 7));
                B[l][j] = (uint8_t)(((rand() & 1) << 7) | ((5 + rand() % 12) << 2) | (rand() & 3));
            }
            for (int i = 0; i < 4; ++i) C[l][i] = trial & 1 ? 37.5f + l : 0.f;
        }
        const int ea = trial < 2 ? -24 : -12, eb = 0;
        uint32_t sa = 0x01010101u * (uint32_t)(127 + ea), sb = 0x01010101u * (uint32_t)(127 + eb);
        void *da, *db; float *dc, *dd;
        hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc((void**)&dc, 1024); hipMalloc((void**)&dd, 1024);
        hipMemcpy(da, A, 2048, hipMemcpyHostToDevice); hipMemcpy(db, B, 2048, hipMemcpyHostToDevice);
        hipMemcpy(dc, C, 1024, hipMemcpyHostToDevice);
        k_mfma8<<<1, 64>>>((const uint32_t*)da, (const uint32_t*)db, dc, dd, sa, sb);
        hipMemcpy(D, dd, 1024, hipMemcpyDeviceToHost);
        double maxrel = 0;
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 4; ++i) {
                const int row = 4 * (l >> 4) + i, col = l & 15;
                double ref = C[l][i];
                for (int q = 0; q < 4; ++q)
                    for (int j = 0; j < 32; ++j)
                        ref += (double)e4m3_val(A[q * 16 + row][j]) * ldexp(1.0, ea) * e5m2_val(B[q * 16 + col][j]);
                double rel = fabs(D[l][i] - ref) / (fabs(ref) + 1e-30);
                if (rel > maxrel) maxrel = rel;
            }
        printf("mfma8 (B in AGPR, C/D VGPR) scaleA 2^%d C=%s: max rel err %.3e %s\n", ea, trial & 1 ? "nonzero" : "0", maxrel,
               maxrel < 1e-5 ? "ok" : "FAIL");
        fails += (trial & 1) && maxrel >= 1e-5;
    }
    // ---- (2) epilogue chain -----------------------------------------------------------------------
    {
        static float T[64][4]; static uint32_t O[64][4];
        srand(5);
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 4; ++i) T[l][i] = ldexpf(1.0f + (float)(rand() % 100000) / 100000.0f, rand() % 9 - 2);
        float* dt; uint32_t* dout;
        hipMalloc((void**)&dt, 1024); hipMalloc((void**)&dout, 1024);
        hipMemcpy(dt, T, 1024, hipMemcpyHostToDevice);
        k_epi<<<1, 64>>>(dt, dout);
        hipMemcpy(O, dout, 1024, hipMemcpyDeviceToHost);
        int bad_h = 0, bad_a = 0, bad_r = 0;
        double worst_r = 0;
        for (int l = 0; l < 64; ++l) {
            _Float16 h[4]; float res[4];
            for (int i = 0; i < 4; ++i) { h[i] = (_Float16)T[l][i]; res[i] = T[l][i] - (float)h[i]; }
            uint16_t hb[4]; memcpy(hb, h, 8);
            if (O[l][0] != (uint32_t)(hb[0] | (hb[1] << 16)) || O[l][1] != (uint32_t)(hb[2] | (hb[3] << 16))) ++bad_h;
            for (int i = 0; i < 4; ++i) {
                const uint8_t ga = (O[l][2] >> (8 * i)) & 255, gr = (O[l][3] >> (8 * i)) & 255;
                const uint8_t wa = f32_to_e5m2_rne(T[l][i]), wr = f32_to_e5m2_rne(res[i]);
                if (ga != wa) { ++bad_a; if (bad_a < 4) printf("  e5m2(t) lane %d i %d: t=%g want 0x%02x got 0x%02x\n", l, i, T[l][i], wa, ga); }
                if (gr != wr && !((gr & 0x7f) == 0 && (wr & 0x7f) == 0)) { ++bad_r; if (bad_r < 4) printf("  resid lane %d i %d: t=%g hi=%g res %g want 0x%02x got 0x%02x\n", l, i, T[l][i], (float)h[i], res[i], wr, gr); }
            }
        }
        printf("epilogue chain: hi pairs %s, e5m2(t) %s, e5m2(t - hi) %s\n", bad_h ? "FAIL" : "ok", bad_a ? "FAIL" : "ok", bad_r ? "FAIL" : "ok");
        fails += (bad_h + bad_a + bad_r) != 0;
    }
    return fails;
}
