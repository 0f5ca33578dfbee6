// Layout probe of the 32x32 MFMA shapes on gfx950: v_mfma_f32_32x32x16_f16 and v_mfma_scale_f32_32x32x64_f8f6f4 (bf6 x bf6).
// Assumed (checked here against a host reference): A lane l = row l%32, k-slot (l/32, j); B lane l = column l%32, k-slot
// (l/32, j); D lane l = column l%32, register r = row 8*(r/4) + 4*(l/32) + r%4; bf6 element e of a lane at bits [6e, 6e+6).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma32_probe.hip -o tools/mfma32_probe && tools/mfma32_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x6 __attribute__((ext_vector_type(6)));

__global__ void k16(const _Float16* A, const _Float16* B, float* D) {   // A [32][16], B [16][32] row-major, D [32][32]
    const int l = threadIdx.x, h = l >> 5, m = l & 31;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = A[m * 16 + 8 * h + j];
        b[j] = B[(8 * h + j) * 32 + m];
    }
    f32x16 d = {0};
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n s_nop 15\n s_nop 15" : "+v"(d) : "v"(a), "v"(b));
    for (int r = 0; r < 16; ++r) D[(8 * (r / 4) + 4 * h + r % 4) * 32 + m] = d[r];
}

__global__ void k6(const uint8_t* A, const uint8_t* B, float* D, int sa, int sb) {   // codes: A [32][64], B [64][32]
    const int l = threadIdx.x, h = l >> 5, m = l & 31;
    uint64_t wa[3] = {0, 0, 0}, wb[3] = {0, 0, 0};
    for (int e = 0; e < 32; ++e) {
        const uint64_t ca = A[m * 64 + 32 * h + e], cb = B[(32 * h + e) * 32 + m];
        const int bit = 6 * e, w = bit >> 6, s = bit & 63;
        wa[w] |= ca << s; wb[w] |= cb << s;
        if (s > 58) { wa[w + 1] |= ca >> (64 - s); wb[w + 1] |= cb >> (64 - s); }
    }
    i32x6 a, b;
    for (int i = 0; i < 3; ++i) {
        a[2 * i] = (int)wa[i]; a[2 * i + 1] = (int)(wa[i] >> 32);
        b[2 * i] = (int)wb[i]; b[2 * i + 1] = (int)(wb[i] >> 32);
    }
    f32x16 d = {0};
    const int va = 0x01010101 * sa, vb = 0x01010101 * sb;
    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:3 blgp:3\n s_nop 15\n s_nop 15"
                 : "+v"(d) : "v"(a), "v"(b), "v"(va), "v"(vb));
    for (int r = 0; r < 16; ++r) D[(8 * (r / 4) + 4 * h + r % 4) * 32 + m] = d[r];
}

// accumulate chain across opcodes, as the body kernel issues it: fp16 MFMAs, then scaled bf6 MFMAs on the SAME accumulator,
// back to back (PAD = 0) or with the pipe drained between all of them (PAD = 1); the results must agree bit for bit
template <int PAD>
__global__ void kchain(const uint32_t* in, float* out) {
    const int l = threadIdx.x;
    f16x8 a[4], b[4];
    i32x6 a6[2], b6[2];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            ((uint32_t*)&a[i])[j] = in[((i * 4 + j) * 64 + l)] & 0x3bff3bffu;          // |x| < 2
            ((uint32_t*)&b[i])[j] = in[((16 + i * 4 + j) * 64 + l)] & 0x3bff3bffu;
        }
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 6; ++j) {
            a6[i][j] = (int)in[((32 + i * 6 + j) * 64 + l)];
            b6[i][j] = (int)in[((44 + i * 6 + j) * 64 + l)];
        }
    f32x16 d;
    for (int r = 0; r < 16; ++r) d[r] = (float)((l * 16 + r) % 7);
    const int sc = 0x01010101 * 122;
#define PADS "s_nop 15\n s_nop 15\n s_nop 15\n"
    if (PAD)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %5, %0\n" PADS "v_mfma_f32_32x32x16_f16 %0, %2, %6, %0\n" PADS
                     "v_mfma_f32_32x32x16_f16 %0, %3, %7, %0\n" PADS "v_mfma_f32_32x32x16_f16 %0, %4, %8, %0\n" PADS
                     "v_mfma_scale_f32_32x32x64_f8f6f4 %0, %9, %11, %0, %13, %13 op_sel_hi:[0,0,0] cbsz:3 blgp:3\n" PADS
                     "v_mfma_scale_f32_32x32x64_f8f6f4 %0, %10, %12, %0, %13, %13 op_sel_hi:[0,0,0] cbsz:3 blgp:3\n" PADS
                     "v_mfma_f32_32x32x16_f16 %0, %1, %6, %0\n" PADS
                     : "+v"(d) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]),
                       "v"(a6[0]), "v"(a6[1]), "v"(b6[0]), "v"(b6[1]), "v"(sc));
    else
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %5, %0\n v_mfma_f32_32x32x16_f16 %0, %2, %6, %0\n"
                     "v_mfma_f32_32x32x16_f16 %0, %3, %7, %0\n v_mfma_f32_32x32x16_f16 %0, %4, %8, %0\n"
                     "v_mfma_scale_f32_32x32x64_f8f6f4 %0, %9, %11, %0, %13, %13 op_sel_hi:[0,0,0] cbsz:3 blgp:3\n"
                     "v_mfma_scale_f32_32x32x64_f8f6f4 %0, %10, %12, %0, %13, %13 op_sel_hi:[0,0,0] cbsz:3 blgp:3\n"
                     "v_mfma_f32_32x32x16_f16 %0, %1, %6, %0\n" PADS
                     : "+v"(d) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]),
                       "v"(a6[0]), "v"(a6[1]), "v"(b6[0]), "v"(b6[1]), "v"(sc));
    for (int r = 0; r < 16; ++r) out[(blockIdx.x * 64 + l) * 16 + r] = d[r];
}

static float bf6(unsigned c) {   // e3m2, bias 3
    const int s = c >> 5, e = (c >> 2) & 7, m = c & 3;
    const float v = e == 0 ? ldexpf((float)m, -4) : ldexpf(1.0f + m / 4.0f, e - 3);
    return s ? -v : v;
}

int main() {
    std::vector<_Float16> A(32 * 16), B(16 * 32);
    std::vector<float> D(32 * 32), ref(32 * 32);
    srand(1);
    for (auto& x : A) x = (_Float16)((rand() % 2001 - 1000) / 500.0f);
    for (auto& x : B) x = (_Float16)((rand() % 2001 - 1000) / 500.0f);
    _Float16 *dA, *dB; float* dD;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    k16<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    double e16 = 0;
    for (int i = 0; i < 32; ++i)
        for (int n = 0; n < 32; ++n) {
            double acc = 0;
            for (int k = 0; k < 16; ++k) acc += (double)A[i * 16 + k] * (double)B[k * 32 + n];
            e16 = fmax(e16, fabs(acc - D[i * 32 + n]));
        }
    printf("v_mfma_f32_32x32x16_f16: max |D - ref| = %.3g (layout %s)\n", e16, e16 < 1e-4 ? "confirmed" : "WRONG");
    std::vector<uint8_t> A6(32 * 64), B6(64 * 32);
    for (auto& x : A6) x = rand() & 63;
    for (auto& x : B6) x = rand() & 63;
    uint8_t *dA6, *dB6;
    hipMalloc(&dA6, A6.size()); hipMalloc(&dB6, B6.size());
    hipMemcpy(dA6, A6.data(), A6.size(), hipMemcpyHostToDevice); hipMemcpy(dB6, B6.data(), B6.size(), hipMemcpyHostToDevice);
    k6<<<1, 64>>>(dA6, dB6, dD, 127 - 3, 127 + 2);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    double e6 = 0, mx = 0;
    for (int i = 0; i < 32; ++i)
        for (int n = 0; n < 32; ++n) {
            double acc = 0;
            for (int k = 0; k < 64; ++k) acc += (double)bf6(A6[i * 64 + k]) * (double)bf6(B6[k * 32 + n]);
            acc *= ldexp(1.0, -3 + 2);
            e6 = fmax(e6, fabs(acc - D[i * 32 + n]));
            mx = fmax(mx, fabs(acc));
        }
    printf("v_mfma_scale_f32_32x32x64_f8f6f4 (bf6 x bf6, scales 2^-3 x 2^2): max |D - ref| = %.3g of %.3g (layout %s)\n", e6, mx,
           e6 < 1e-3 * mx ? "confirmed" : "WRONG");
    // accumulate chain across opcodes
    const int NB = 2048;
    std::vector<uint32_t> in(56 * 64);
    for (auto& x : in) x = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
    uint32_t* din; float *o0, *o1;
    hipMalloc(&din, in.size() * 4); hipMalloc(&o0, NB * 64 * 16 * 4); hipMalloc(&o1, NB * 64 * 16 * 4);
    hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    kchain<0><<<NB, 64>>>(din, o0);
    kchain<1><<<NB, 64>>>(din, o1);
    std::vector<uint32_t> h0(NB * 64 * 16), h1(NB * 64 * 16);
    hipMemcpy(h0.data(), o0, h0.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), o1, h1.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < h0.size(); ++i) bad += h0[i] != h1[i];
    printf("accumulate chain f16 x4 -> bf6 x2 -> f16 on one accumulator, back to back vs drained: %zu of %zu words differ\n", bad, h0.size());
    return 0;
}
