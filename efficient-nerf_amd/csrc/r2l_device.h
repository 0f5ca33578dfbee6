// Device-side building blocks shared by the R2L and NeRF-teacher MLP kernels: fp16 MFMA
// k-step, LDS weight ring (LDS-DMA + counted vmcnt + one barrier per chunk), activation
// hi/lo split, exact-argument sin/cos for the sinusoidal embeddings.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "r2l_common.h"

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

#define AS1(p) ((const __attribute__((address_space(1))) void*)(p))
#define AS3(p) ((__attribute__((address_space(3))) void*)(p))

extern __shared__ __attribute__((aligned(16))) char smem[];

// ------------------------------------------------------------------------------------
// sin / cos of 2^l * x, l = 0..9, matching sin/cos of the exactly scaled fp32 argument.
// x/(2 pi) is kept as an unevaluated sum rh + rl (two-float), scaled by the exact power
// of two, the integer part is removed exactly (v_rndne), and sin(2 pi g) is evaluated on
// the folded fraction |g'| <= 1/4 by v_sin_f32.
// ------------------------------------------------------------------------------------
#define R2L_INV2PI_HI 0.15915494f              // fl32(1/(2 pi))
#define R2L_INV2PI_LO 6.4206382e-09f           // 1/(2 pi) - fl32(1/(2 pi))  (set by host check)
#define R2L_2PI 6.2831855f

struct Rev {  // x / (2 pi) = rh + rl
    float rh, rl;
};

__device__ __forceinline__ Rev to_rev(float x) {
    Rev r;
    r.rh = x * R2L_INV2PI_HI;
    r.rl = fmaf(x, R2L_INV2PI_LO, fmaf(x, R2L_INV2PI_HI, -r.rh));
    return r;
}

// sin (is_cos = false) or cos (true) of x * 2^l given x/(2 pi) = rh + rl, pow2l = 2^l.
// rh*2^l is exact, t - rint(t) is exact, so g = frac(x*2^l/(2 pi)) in [-1/2, 1/2] keeps
// ~2^-26 absolute accuracy; cos uses cos(2 pi g) = sin(2 pi (1/4 - |g|)).  The turn fraction
// goes to v_sin_f32 (sin(2 pi x), |err| <= 1.3e-7 on [-1, 1], measured: tools/vsin_test.hip).
__device__ __forceinline__ float trig_pow2(Rev r, float pow2l, bool is_cos) {
    float t = r.rh * pow2l;
    float u = t - rintf(t);
    float g = fmaf(r.rl, pow2l, u);                  // frac(x*2^l/(2 pi)) in [-1/2, 1/2]
    float arg = is_cos ? (0.25f - fabsf(g)) : g;     // cos(2 pi g) = sin(2 pi (1/4 - |g|)), exact fold
    return __builtin_amdgcn_sinf(arg);               // v_sin_f32 reduces |arg| <= 1/2 itself
}

// ------------------------------------------------------------------------------------
// LDS weight ring + MFMA k-step
// ------------------------------------------------------------------------------------
template <int NP>
struct KCfg {
    static constexpr int AUX = R2L_FRAGS * NP * R2L_FRAG_BYTES;  // aux offset in a chunk
    static constexpr int CH = AUX + R2L_AUX_BYTES;               // chunk bytes
    static constexpr int NBUF = (NP == 2) ? 4 : 6;               // LDS ring slots
    static constexpr int D = NBUF - 1;                           // chunks issued ahead
    static constexpr int P = 4 * NP + 1;                         // LDS-DMA ops / wave / chunk
    static constexpr int WAIT_MID = (D - 2) * P;                 // certify chunk c+1 at mid-c
    static constexpr int WAIT_PRO = (D - 1) * P;                 // prologue: certify chunk 0
    static constexpr int LDS = NBUF * CH;
};

template <int NP>
struct AFrag {  // one A (weight) fragment: hi [, lo]
    f16x8 h, l;
};

template <int NP>
struct Ring {
    const char* wimg;   // chunk stream of one ray tile (periodic)
    int cpt;            // chunks per tile
    int issue_pos;      // next chunk (position in the tile image) to issue
    uint32_t issue_off; // LDS byte offset of the slot it goes to
    uint32_t use_off;   // LDS byte offset of the chunk being consumed
    int wave, lane;
    AFrag<NP> pre;      // fragment 0 of the chunk at use_off, already read from LDS
};

// LDS-DMA of one chunk: per wave 4*NP pieces of 1 KiB (`global_load_lds_dwordx4`, 16 B/lane) +
// one 256 B piece of the aux block (`global_load_lds_dword`).  The instruction's immediate
// offset is added to the global AND the LDS address, so one (global, LDS) base pair serves four
// consecutive pieces.  MUBUF `... lds` loads are not usable here: their LDS base is M0[15:0]
// and the ring spans > 64 KiB.
template <int NP>
__device__ __forceinline__ void ring_issue_piece(Ring<NP>& R, int piece) {
    typedef KCfg<NP> C;
    const char* src = R.wimg + (size_t)R.issue_pos * C::CH + R.wave * (4 * NP * R2L_FRAG_BYTES) + (uint32_t)R.lane * 16u;
    const uint32_t dst = R.issue_off + R.wave * (4 * NP * R2L_FRAG_BYTES);
    const int last = (NP == 2) ? 7 : 4;
    switch (piece) {
        case 0: __builtin_amdgcn_global_load_lds(AS1(src), AS3(smem + dst), 16, 0, 0); break;
        case 1: __builtin_amdgcn_global_load_lds(AS1(src), AS3(smem + dst), 16, 1024, 0); break;
        case 2: __builtin_amdgcn_global_load_lds(AS1(src), AS3(smem + dst), 16, 2048, 0); break;
        case 3: __builtin_amdgcn_global_load_lds(AS1(src), AS3(smem + dst), 16, 3072, 0); break;
        case 4: if (NP == 2) __builtin_amdgcn_global_load_lds(AS1(src + 4096), AS3(smem + dst + 4096), 16, 0, 0); break;
        case 5: if (NP == 2) __builtin_amdgcn_global_load_lds(AS1(src + 4096), AS3(smem + dst + 4096), 16, 1024, 0); break;
        case 6: if (NP == 2) __builtin_amdgcn_global_load_lds(AS1(src + 4096), AS3(smem + dst + 4096), 16, 2048, 0); break;
        case 7: if (NP == 2) __builtin_amdgcn_global_load_lds(AS1(src + 4096), AS3(smem + dst + 4096), 16, 3072, 0); break;
        default: break;
    }
    if (piece == last) {  // the aux block rides with the last piece, then the ring advances
        const char* asrc = R.wimg + (size_t)R.issue_pos * C::CH + C::AUX + R.wave * 256;
        __builtin_amdgcn_global_load_lds(AS1(asrc + (uint32_t)R.lane * 4u), AS3(smem + R.issue_off + C::AUX + R.wave * 256),
                                         4, 0, 0);
        R.issue_pos = (R.issue_pos + 1 == R.cpt) ? 0 : R.issue_pos + 1;
        R.issue_off = (R.issue_off + C::CH == (uint32_t)C::LDS) ? 0u : R.issue_off + C::CH;
    }
}

template <int NP>
__device__ __forceinline__ void ring_issue(Ring<NP>& R) {  // all pieces at once (prologue, chunk tails)
#pragma unroll
    for (int piece = 0; piece < 8; ++piece) ring_issue_piece<NP>(R, piece);
}

#define R2L_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// Mid-chunk rendezvous while consuming chunk c: every wave's LDS-DMA of chunk c+1 has
// landed (counted vmcnt, then the barrier), every wave is past chunk c-1, so its slot may be
// refilled with chunk c+D: by ring_issue_piece over the following k-steps, or at once (ring_mid).
template <int NP>
__device__ __forceinline__ void ring_sync(Ring<NP>& R) {
    R2L_WAIT_VMCNT(KCfg<NP>::WAIT_MID);
    __builtin_amdgcn_s_barrier();
}

// fragment position f (0..15) inside a chunk: rendezvous + refill at 8.  (Spreading the
// refill over positions 8..15, one piece per k-step, measured 8 % SLOWER than the burst.)
// (Staggering the refill burst by wave -- positions 8, 10, 12, 14 -- changes nothing: the issue cost is
// per wave, not contention between the four waves.)
template <int NP>
__device__ __forceinline__ void ring_step(Ring<NP>& R, int f) {
    if (f == R2L_FRAGS / 2) {
        ring_sync<NP>(R);
        ring_issue<NP>(R);
    }
}

template <int NP>
__device__ __forceinline__ void ring_mid(Ring<NP>& R) {
    ring_sync<NP>(R);
    ring_issue<NP>(R);
}

template <int NP>
__device__ __forceinline__ uint32_t ring_next_off(uint32_t off) {
    typedef KCfg<NP> C;
    return (off + C::CH == (uint32_t)C::LDS) ? 0u : off + C::CH;
}

template <int NP>
__device__ __forceinline__ void ring_next(Ring<NP>& R) {
    R.use_off = ring_next_off<NP>(R.use_off);
}

template <int NP>
__device__ __forceinline__ AFrag<NP> read_frag(uint32_t lane_base, int frag) {
    AFrag<NP> a;
    a.h = *reinterpret_cast<const f16x8*>(smem + lane_base + (frag * NP) * R2L_FRAG_BYTES);
    if (NP == 2) a.l = *reinterpret_cast<const f16x8*>(smem + lane_base + (frag * NP + 1) * R2L_FRAG_BYTES);
    return a;
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

// v_mfma_scale_f32_16x16x128_f8f6f4: A = weights e4m3 (cbsz 0), B = activations e5m2 (blgp 1); the E8M0
// scale of A (lane-uniform, byte 0) undoes the host's power-of-two shift of the packed weights.
#define MFMA8(a, b, c, scale_a) \
    __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4((a), (b), (c), 0, 1, 0, (scale_a), 0, 0x7f7f7f7f)
#define R2L_MIX_SCALE_WL (0x01010101 * (127 - R2L_MIX_WL_SHIFT))  // products with (w - hi) * 2^7
#define R2L_MIX_SCALE_W (0x01010101 * (127 + R2L_MIX_W_SHIFT))    // products with w * 2^-5

// one k-step (32 inputs) on one 16x16 output tile: ah*bh [+ ah*bl + al*bh]
template <int NP>
__device__ __forceinline__ f32x4 mfma_step(const AFrag<NP>& a, const f16x8& bh, const f16x8& bl, f32x4 acc) {
    acc = MFMA(a.h, bh, acc);
    if (NP == 2) {
        acc = MFMA(a.h, bl, acc);
        acc = MFMA(a.l, bh, acc);
    }
    return acc;
}

// split an fp32 activation (already multiplied by act_scale) into fp16 hi (+ lo)
template <int NP>
__device__ __forceinline__ void split_store(float a, f16x8& hi, f16x8& lo, int j) {
    f16 h = (f16)a;
    hi[j] = h;
    if (NP == 2) lo[j] = (f16)fmaf((float)h, -1.0f, a);  // a - hi, fma-shaped so it can become v_fma_mix*_f16
}

// ---- pairwise forms (two consecutive accumulator registers -> one dword of a fragment) ----
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f16x2 pack_hi(float v0, float v1) {  // v_cvt_pk_f16_f32 (round to nearest even)
    return __builtin_convertvector(f32x2{v0, v1}, f16x2);
}
// Residuals v - (float)hi are written fma((float)hi, neg1, v) with neg1 = -1.0f arriving as a kernel
// argument: a compile-time -1 folds into cvt + sub (2 VALU), the opaque multiplier selects
// v_fma_mix_f32 / v_fma_mixlo|hi_f16, which read the f16 half directly (1 VALU) and stay exact.
__device__ __forceinline__ uint32_t pack_lo(f16x2 h, float v0, float v1, float neg1) {
    const f16x2 l = {(f16)fmaf((float)h[0], neg1, v0), (f16)fmaf((float)h[1], neg1, v1)};
    return __builtin_bit_cast(uint32_t, l);
}
__device__ __forceinline__ f16x2 get_pair(const f16x8& f, int idx) { return f16x2{f[2 * idx], f[2 * idx + 1]}; }
__device__ __forceinline__ void set_dword(f16x8& f, int idx, uint32_t w) {
    i32x4 t = __builtin_bit_cast(i32x4, f);
    t[idx] = (int)w;
    f = __builtin_bit_cast(f16x8, t);
}
// split two activations into dword `idx` of the hi (and lo) fragment
template <int NP>
__device__ __forceinline__ void split_store2(float v0, float v1, f16x8& hi, f16x8& lo, int idx, float neg1) {
    const f16x2 h = pack_hi(v0, v1);
    set_dword(hi, idx, __builtin_bit_cast(uint32_t, h));
    if (NP == 2) set_dword(lo, idx, pack_lo(h, v0, v1, neg1));
}

// accumulator init of one row tile = aux[feat_off + 4q + i], i = 0..3 (bias pre-multiplied by
// the layer scale); the same for both column tiles
template <int NP>
__device__ __forceinline__ f32x4 acc_init(uint32_t slot_off, int feat_off, int q) {
    return *reinterpret_cast<const f32x4*>(smem + slot_off + KCfg<NP>::AUX + (feat_off + 4 * q) * 4);
}

template <int NP>
__device__ __forceinline__ float aux_inv_scale(uint32_t slot_off) {
    return *reinterpret_cast<const float*>(smem + slot_off + KCfg<NP>::AUX + 32 * 4);
}

