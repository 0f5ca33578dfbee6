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
