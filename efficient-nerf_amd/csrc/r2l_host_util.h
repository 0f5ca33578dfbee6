// Host helpers shared by r2l_capi.hip and nerf_capi.hip.
#pragma once
#include <stddef.h>

int r2l_set_error(int code, const char* fmt, ...);
int r2l_require_gfx950(int* n_cu);
void r2l_linspace01(int steps, float* out);                        // torch.linspace(0,1,steps), f32
void r2l_z_vals(int steps, float near_, float far_, float* out);  // near*(1-t)+far*t, f32 per-op rounding
float r2l_pow2_scale(const float* w, size_t n);                    // 2^e with max|w|*2^e in [2^12,2^13)
void r2l_split_f16(float v, _Float16* hi, _Float16* lo);
unsigned char r2l_f32_to_e4m3(float v);
int r2l_layer_exponent(const float* w, size_t n);                  // e with max|w| in [2^(e-1), 2^e); -4 for all-zero
unsigned r2l_f_to_bf6(double v);                                   // OCP bf6 = e3m2 code, round-to-nearest-even, saturating                            // OCP e4m3fn, round-to-nearest-even, saturating

// Persistent workgroups take ray tiles b, b + grid, ...: with one workgroup per CU the launch lasts ceil(n_tiles / n_cu) tile
// times and the last round is partly empty (5,000 tiles on 256 CUs: 19.5 rounds).  The smallest grid with the same number of
// rounds gives every workgroup the same work and leaves the idle CUs' share of the package power to the others: 250 workgroups
// for 5,000 tiles, -1.3 % kernel time (256 -> 250 same-call A/B; 228 = 22 rounds +0.9 %, 200 = 25 rounds +3.9 %).
#ifdef R2L_GRID_OVERRIDE      // experiment (tools/build_variant.sh CAPI_DEF=-DR2L_GRID_OVERRIDE=250)
static inline int balanced_grid(int n_tiles, int) { return n_tiles < R2L_GRID_OVERRIDE ? n_tiles : R2L_GRID_OVERRIDE; }
#else
static inline int balanced_grid(int n_tiles, int n_cu) {
    if (n_tiles <= n_cu) return n_tiles;
    const int rounds = (n_tiles + n_cu - 1) / n_cu;
    return (n_tiles + rounds - 1) / rounds;
}
#endif
