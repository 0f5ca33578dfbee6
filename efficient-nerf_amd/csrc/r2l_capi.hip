// C-ABI host side for the R2L student: context, host-side weight re-packing into the
// MFMA-native chunk stream (r2l_common.h), launches.  Declared in include/r2l_hip.h.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/r2l_hip.h"
#include "r2l_common.h"
#include "r2l_kernels.h"
#include "r2l_host_util.h"

static thread_local std::string g_err;

const char* r2l_last_error(void) { return g_err.c_str(); }

int r2l_set_error(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

int r2l_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, i) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

int r2l_require_gfx950(int* n_cu) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return r2l_set_error(R2L_ENOGPU, "hipGetDevice: %s", hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return r2l_set_error(R2L_ENOGPU, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return r2l_set_error(R2L_ENOGPU, "device %d is %s; this library is built for gfx950 only", dev,
                             prop.gcnArchName);
    if (n_cu) *n_cu = prop.multiProcessorCount;
    return R2L_OK;
}

// torch.linspace(0, 1, steps) on CPU for float32 (aten RangeFactories: symmetric fill)
void r2l_linspace01(int steps, float* out) {
    volatile float step = (1.0f - 0.0f) / (float)(steps - 1);
    int halfway = steps / 2;
    for (int i = 0; i < steps; ++i) {
        volatile float v;
        if (i < halfway) {
            volatile float m = step * (float)i;
            v = 0.0f + m;
        } else {
            volatile float m = step * (float)(steps - i - 1);
            v = 1.0f - m;
        }
        out[i] = v;
    }
}

// near * (1 - t) + far * t in float32, one rounding per op (model/nerf_raybased.py:90)
void r2l_z_vals(int steps, float near_, float far_, float* out) {
    std::vector<float> t(steps);
    r2l_linspace01(steps, t.data());
    for (int i = 0; i < steps; ++i) {
        volatile float a = 1.0f - t[i];
        volatile float b = near_ * a;
        volatile float c = far_ * t[i];
        volatile float z = b + c;
        out[i] = z;
    }
}

// power-of-two scale that brings max|w| into [2^12, 2^13)
float r2l_pow2_scale(const float* w, size_t n) {
    float m = 0.f;
    for (size_t i = 0; i < n; ++i) {
        float a = fabsf(w[i]);
        if (a > m && isfinite(a)) m = a;
    }
    if (m == 0.f) return 1.0f;
    int e;
    frexpf(m, &e);  // m = f * 2^e, f in [0.5,1)
    return ldexpf(1.0f, 13 - e);
}

void r2l_split_f16(float v, _Float16* hi, _Float16* lo) {
    _Float16 h = (_Float16)v;
    *hi = h;
    if (lo) *lo = (_Float16)(v - (float)h);
}

// OCP e4m3fn (bias 7, 3 mantissa bits, max 448, no inf), round to nearest even, saturating
unsigned char r2l_f32_to_e4m3(float v) {
    const unsigned char sgn = signbit(v) ? 0x80 : 0;
    const float a = fabsf(v);
    if (!(a == a)) return 0x7f;
    if (a >= 464.0f) return sgn | 0x7e;
    int e;
    frexpf(a, &e);
    const int E = e - 1;  // a = 1.x * 2^E
    if (a == 0.f || E < -6) return sgn | (unsigned char)nearbyintf(ldexpf(a, 9));  // subnormal step 2^-9 (8 -> min normal)
    int qn = (int)nearbyintf(ldexpf(a, 3 - E));  // 8..16
    int EE = E;
    if (qn == 16) {
        qn = 8;
        ++EE;
    }
    return sgn | (unsigned char)(((EE + 7) << 3) | (qn - 8));
}

#define R2L_GUARD_PERIOD_DEFAULT 8   // every 8th body launch runs the range-guard build (and the first after a weight load)
#define R2L_CALIB_MAX_CALLS 8  // a measurement that only ever sees thin calls (< 1,024 rays) is closed after this many

// OCP e4m3fn from a double: round to nearest even, saturating at 448 (what v_cvt_scalef32_pk_fp8_f16 does under
// MODE.FP16_OVFL, tools/fp8_probe.hip)
static unsigned char r2l_f_to_e4m3(double v) {
    const unsigned char sgn = signbit(v) ? 0x80 : 0;
    const double a = fabs(v);
    if (!(a == a)) return 0x7f;
    if (a >= 448.0) return sgn | 0x7e;
    int e;
    frexp(a, &e);
    int E = e - 1;                                         // a = 1.x * 2^E
    if (a == 0.0 || E < -6) return sgn | (unsigned char)nearbyint(ldexp(a, 9));   // subnormal step 2^-9 (8 -> min normal)
    int qn = (int)nearbyint(ldexp(a, 3 - E));              // 8..16
    if (qn == 16) {
        qn = 8;
        ++E;
    }
    if (E > 8 || (E == 8 && qn > 14)) return sgn | 0x7e;
    return sgn | (unsigned char)(((E + 7) << 3) | (qn - 8));
}

struct r2l_ctx {
    int H, W, n_block, use_residual, mode;
    double focal;
    float near_, far_;
    float z[16];
    float act_scale;
    int n_cu;
    bool loaded;
    std::vector<std::vector<float>> host_w;  // state_dict order
    char* d_img[7];                           // [mode] packed image (FP16_FP8: the 32 head chunks, hi|lo layout)
    size_t img_bytes[7];
    char* d_body;                             // split modes: body stream v3 (r2l_body.hip) | aux blocks | tail
    int body_mode = -1;                       // ... of this mode (bf6 or e4m3 terms: chunk geometry and operand codes differ)
    size_t body_bytes, aux_off, tail_off;
    // R2L_PREC_FP16_SPLIT / _SPLIT8: blocks [0, split_block) run on the three-pass stream d_body2, blocks [split_block, n_block) on d_body
    // (bf6 / e4m3 terms)
    char* d_body2 = nullptr;
    size_t aux_split_off = 0;
    size_t body2_bytes = 0, aux_off2 = 0, tail_off2 = 0;
    int split_block = -1;                     // -1: n_block / 2 until r2l_set_split_block
    std::vector<int> act;                     // FP16_FP8: 2 n_block + 1 activation exponents (packed into the aux blocks)
    // activations as slopes (act(v) = max(v, s v): 0 relu, 0.01 LeakyReLU, 1 none): head, inside a block, behind a block.  The
    // generated kernels are built for (0, 0, 1); anything else runs in the compiler-scheduled modes only (r2l_set_activations)
    float act_head = 0.0f, act_in = 0.0f, act_out = 1.0f;
    int block_resid = 1;                      // 0: trial.body_arch = mlp (pairs of plain Linear + act layers, no residual inside the pair)
    int fuse_tail = 1;                        // FP16_FP8 with the global skip: rgb written by the body kernel's fused tail
    int calib_pending = 0;                    // the next FP16_FP8 render derives them from its own head output (device side)
    int calib_started = 0;                    // maxima of earlier, smaller calls are in d_stats: the next measurement adds to them
    int calib_calls = 0;                      // thin calls the open measurement has seen (closed after R2L_CALIB_MAX_CALLS)
    // range tracking (r2l_get_range_status), ONE allocation so that the host reads it with one copy:
    // d_range[0] = max h0 of EVERY ray since the last reset (head launch), [1..3] spare;
    // d_gstats = d_range + 4: [2 n_block + 1] maxima of every operand set over every ray of the guarded body launches since
    // the last reset; d_exps = d_gstats + 2 n_block + 1: [2 n_block + 1] the activation exponents in use (what the aux blocks
    // hold, written beside them by the calibration kernels and by r2l_set_act_exponents)
    unsigned* d_range = nullptr;
    unsigned* d_gstats = nullptr;
    int* d_exps = nullptr;
    int guard_period = R2L_GUARD_PERIOD_DEFAULT;
    long long n_body = 0, n_guarded = 0;      // FP16_FP8 body launches since the last reset / of them guarded
    long long n_since_load = 0;               // ... since r2l_load_weights (the guard's phase)
    hipStream_t last_stream = nullptr;        // stream of the newest render: the synchronous host copies wait for it
    float* d_wcal = nullptr;                  // fp32 W1^T | b1' | W2^T per block for the calibration kernel
    unsigned* d_stats = nullptr;
    float* d_xa;                              // FP16_FP8: head output of one launch slice (1 KiB per ray)
    float* d_xb;                              // body output: only the unfused form (no global skip, r2l_debug_set_fused_tail(0))
    int x_tiles, xb_tiles;                    // capacity of d_xa / d_xb in ray tiles
    float* d_scratch;
    float* d_z;  // device copy of z
    bool timing;
    std::vector<hipEvent_t> ev;  // pairs
    int ev_used;
};

static int np_of(int mode) { return mode == R2L_PREC_FP16X1 ? 1 : 2; }
enum { R2L_STREAM_BF6 = 0, R2L_STREAM_E4M3 = 1, R2L_STREAM_BF6R = 2, R2L_STREAM_F16 = 3 };   // body stream layouts (pack_body_v3)
static bool mode_ok(int mode) { return mode >= R2L_PREC_FP16X3 && mode <= R2L_PREC_FP16_SPLIT8; }
// the split rungs: head + leading blocks in three passes, the rest with bf6 (SPLIT) or e4m3 (SPLIT8) terms
static bool two_part(int mode) { return mode == R2L_PREC_FP16_SPLIT || mode == R2L_PREC_FP16_SPLIT8; }
// ... the modes whose correction terms are e4m3 (operand top 448, 32 KiB chunks)
static bool e4m3_terms(int mode) { return mode == R2L_PREC_FP16_E4M3 || mode == R2L_PREC_FP16_SPLIT8; }
// the modes with the generated head launch + generated body kernel
static bool split_mode(int mode) {
    return mode == R2L_PREC_FP16_FP8 || mode == R2L_PREC_FP16_E4M3 || mode == R2L_PREC_FP16X3_ASM || two_part(mode);
}
// ... of them those with low-precision correction terms in the body: calibrated operand scales, range tracking
static bool scaled_mode(int mode) { return mode == R2L_PREC_FP16_FP8 || mode == R2L_PREC_FP16_E4M3 || two_part(mode); }
// the head launch in three fp16 passes (r2l_headx_kernel)?  FP16_SPLIT too: on a trained network the bf6-term head alone costs
// 9e-5 of the 1e-4 contract (profiles/r05_split_time.txt) at 4 % of the MACs
static bool head_x3(int mode) { return mode == R2L_PREC_FP16X3_ASM || two_part(mode); }
static int stream_of(int mode) {
    return e4m3_terms(mode) ? R2L_STREAM_E4M3 : mode == R2L_PREC_FP16X3_ASM ? R2L_STREAM_F16 :
           (R2L_BF6_CHUNK == 28672 ? R2L_STREAM_BF6 : R2L_STREAM_BF6R);
}
#define R2L_N_MODES 7


#ifndef R2L_SLICE_TILES
#define R2L_SLICE_TILES 8192   // FP16_FP8: ray tiles per head / body launch pair (1 KiB of h0 per ray)
#endif


int r2l_create(r2l_ctx** out, int H, int W, double focal, float near_, float far_, int n_sample, int L,
               int width, int n_block, int use_residual, int precision_mode) {
    if (!out) return r2l_set_error(R2L_EINVAL, "out is NULL");
    *out = nullptr;
    if (n_sample != R2L_NSAMPLE || L != R2L_L || width != R2L_WIDTH)
        return r2l_set_error(R2L_EINVAL,
                             "unsupported R2L shape n_sample=%d L=%d width=%d (built for n_sample=16, multires=10, "
                             "netwidth=256)",
                             n_sample, L, width);
    if (H <= 0 || W <= 0 || n_block < 0 || !(focal > 0))
        return r2l_set_error(R2L_EINVAL, "bad geometry H=%d W=%d focal=%g n_block=%d", H, W, focal, n_block);
    if (!mode_ok(precision_mode)) return r2l_set_error(R2L_EINVAL, "bad precision_mode %d", precision_mode);
    int n_cu = 0;
    int rc = r2l_require_gfx950(&n_cu);
    if (rc) return rc;
    r2l_ctx* c = new r2l_ctx();
    c->H = H;
    c->W = W;
    c->focal = focal;
    c->near_ = near_;
    c->far_ = far_;
    c->n_block = n_block;
    c->use_residual = use_residual ? 1 : 0;
    c->mode = precision_mode;
    c->act_scale = 16.0f;
    c->n_cu = n_cu;
    c->loaded = false;
    for (int m = 0; m < R2L_N_MODES; ++m) {
        c->d_img[m] = nullptr;
        c->img_bytes[m] = 0;
    }
    c->d_body = nullptr;
    c->body_bytes = c->aux_off = c->tail_off = 0;
    c->d_xa = c->d_xb = nullptr;
    c->x_tiles = c->xb_tiles = 0;
    c->d_scratch = nullptr;
    c->d_z = nullptr;
    c->timing = false;
    c->ev_used = 0;
    r2l_z_vals(R2L_NSAMPLE, near_, far_, c->z);
    size_t scr = (size_t)n_cu * R2L_WAVES * 32 * 256 * sizeof(float);
    hipError_t e = hipMalloc((void**)&c->d_scratch, scr);
    if (e != hipSuccess) {
        delete c;
        return r2l_set_error(R2L_EHIP, "hipMalloc scratch: %s", hipGetErrorString(e));
    }
    e = hipMalloc((void**)&c->d_z, sizeof c->z);
    if (e == hipSuccess) e = hipMemcpy(c->d_z, c->z, sizeof c->z, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        r2l_destroy(c);
        return r2l_set_error(R2L_EHIP, "hipMalloc/hipMemcpy z_vals: %s", hipGetErrorString(e));
    }
    *out = c;
    return R2L_OK;
}

void r2l_destroy(r2l_ctx* c) {
    if (!c) return;
    for (int m = 0; m < R2L_N_MODES; ++m)
        if (c->d_img[m]) (void)hipFree(c->d_img[m]);
    if (c->d_body) (void)hipFree(c->d_body);
    if (c->d_body2) (void)hipFree(c->d_body2);
    if (c->d_xa) (void)hipFree(c->d_xa);
    if (c->d_xb) (void)hipFree(c->d_xb);
    if (c->d_wcal) (void)hipFree(c->d_wcal);
    if (c->d_stats) (void)hipFree(c->d_stats);
    if (c->d_range) (void)hipFree(c->d_range);   // d_gstats, d_exps: the same allocation
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    if (c->d_z) (void)hipFree(c->d_z);
    for (auto& e : c->ev) (void)hipEventDestroy(e);
    delete c;
}

// ---- packing ---------------------------------------------------------------------------
static void put_frag(char* chunk, int np, int frag, int lane, int j, float v) {
    _Float16 hi, lo;
    r2l_split_f16(v, &hi, np == 2 ? &lo : nullptr);
    _Float16* ph = reinterpret_cast<_Float16*>(chunk + (size_t)(frag * np + 0) * R2L_FRAG_BYTES + lane * 16);
    ph[j] = hi;
    if (np == 2) {
        _Float16* pl = reinterpret_cast<_Float16*>(chunk + (size_t)(frag * np + 1) * R2L_FRAG_BYTES + lane * 16);
        pl[j] = lo;
    }
}

static void pack_image_host(const r2l_ctx* c, int mode, std::vector<char>& img);
static int ensure_x(r2l_ctx* c, int tiles, bool need_xb);
static int pack_body_v3(const r2l_ctx* c, int e4m3, std::vector<char>& out, size_t* aux_off, size_t* tail_off);
static int pack_head_v1(const r2l_ctx* c, std::vector<char>& out, int f16 = 0);

static bool default_acts(const r2l_ctx* c) { return c->act_head == 0.0f && c->act_in == 0.0f && c->act_out == 1.0f && c->block_resid == 1; }

static int build_image(r2l_ctx* c, int mode) {
    std::vector<char> img;
    if (split_mode(mode) && !default_acts(c))
        return r2l_set_error(R2L_EINVAL, "activation slopes head %g / inner %g / out %g: the generated kernels (fp16_fp8, fp16_e4m3, "
                                         "fp16x3_asm) are built for relu / relu / none (act=relu, trial.inact=relu, trial.outact=none); "
                                         "fp16x3 renders the others", c->act_head, c->act_in, c->act_out);
    if (split_mode(mode)) {
        // head launch: the stream of r2l_head_kernel (bf6 terms in both modes); body + tail: the v3 stream of the mode
        int rc = pack_head_v1(c, img, head_x3(mode));
        if (rc) return rc;
        std::vector<char> body;
        rc = pack_body_v3(c, stream_of(mode), body, &c->aux_off, &c->tail_off);
        if (rc) return rc;
        c->body_mode = mode;
        if (c->d_body) {
            (void)hipFree(c->d_body);
            c->d_body = nullptr;
        }
        hipError_t eb = hipMalloc((void**)&c->d_body, body.size());
        if (eb != hipSuccess) return r2l_set_error(R2L_EHIP, "hipMalloc body stream: %s", hipGetErrorString(eb));
        eb = hipMemcpy(c->d_body, body.data(), body.size(), hipMemcpyHostToDevice);
        if (eb != hipSuccess) return r2l_set_error(R2L_EHIP, "hipMemcpy body stream: %s", hipGetErrorString(eb));
        c->body_bytes = body.size();
        if (c->d_body2) {
            (void)hipFree(c->d_body2);
            c->d_body2 = nullptr;
        }
        if (two_part(mode) && c->n_block > 0) {     // ... and the three-pass stream for the blocks in front of the split
            std::vector<char> body2;
            rc = pack_body_v3(c, R2L_STREAM_F16, body2, &c->aux_off2, &c->tail_off2);
            if (rc) return rc;
            // behind it: room for the aux blocks of a truncated bf6 launch (r2l_launch_split_aux, per slice)
            c->aux_split_off = body2.size();
            eb = hipMalloc((void**)&c->d_body2, body2.size() + (size_t)c->n_block * R2L_BODY_AUX_BYTES);
            if (eb == hipSuccess) eb = hipMemcpy(c->d_body2, body2.data(), body2.size(), hipMemcpyHostToDevice);
            if (eb != hipSuccess) return r2l_set_error(R2L_EHIP, "three-pass body stream: %s", hipGetErrorString(eb));
            c->body2_bytes = body2.size();
        }
        // calibration operands: per block W1^T | b1' (the folded bias, real units) | W2^T in fp32
        if (c->n_block > 0) {
            const size_t per = 2 * 65536 + 256;
            std::vector<float> wc((size_t)c->n_block * per);
            for (int b = 0; b < c->n_block; ++b) {
                const float* W1 = c->host_w[2 + 4 * b].data();
                const float* W2 = c->host_w[4 + 4 * b].data();
                const float* auxb = reinterpret_cast<const float*>(body.data() + c->aux_off + (size_t)b * R2L_BODY_AUX_BYTES);
                float* o = wc.data() + (size_t)b * per;
                for (int k = 0; k < 256; ++k)
                    for (int f = 0; f < 256; ++f) {
                        o[k * 256 + f] = W1[(size_t)f * 256 + k];
                        o[65536 + 256 + k * 256 + f] = W2[(size_t)f * 256 + k];
                    }
                for (int f = 0; f < 256; ++f) o[65536 + f] = auxb[f] / c->act_scale;
            }
            if (c->d_wcal) (void)hipFree(c->d_wcal);
            if (c->d_stats) (void)hipFree(c->d_stats);
            c->d_wcal = nullptr;
            c->d_stats = nullptr;
            eb = hipMalloc((void**)&c->d_wcal, wc.size() * sizeof(float));
            if (eb == hipSuccess) eb = hipMemcpy(c->d_wcal, wc.data(), wc.size() * sizeof(float), hipMemcpyHostToDevice);
            if (eb == hipSuccess) eb = hipMalloc((void**)&c->d_stats, (size_t)(2 * c->n_block + 1) * sizeof(unsigned));
            if (eb != hipSuccess) return r2l_set_error(R2L_EHIP, "calibration operands: %s", hipGetErrorString(eb));
        }
        // range tracking words; the h0 buffer of the context's own frame: nothing is allocated inside a render of <= H x W rays
        const size_t nset1 = (size_t)2 * c->n_block + 1;
        if (!c->d_range) {
            eb = hipMalloc((void**)&c->d_range, (4 + 2 * nset1) * sizeof(unsigned));
            if (eb == hipSuccess) {
                c->d_gstats = c->d_range + 4;
                c->d_exps = reinterpret_cast<int*>(c->d_gstats + nset1);
            }
        }
        if (eb == hipSuccess) eb = hipMemset(c->d_range, 0, (4 + nset1) * sizeof(unsigned));
        if (eb == hipSuccess) {      // the exponents pack_body_v3 has just put into the aux blocks
            std::vector<int> ex(nset1);
            for (size_t j = 0; j < nset1; ++j) ex[j] = j < c->act.size() ? c->act[j] : R2L_ACT_EXP;
            if (c->n_block > 0) ex[nset1 - 1] = ex[0];
            eb = hipMemcpy(c->d_exps, ex.data(), nset1 * sizeof(int), hipMemcpyHostToDevice);
        }
        if (eb != hipSuccess) return r2l_set_error(R2L_EHIP, "range tracking words: %s", hipGetErrorString(eb));
        c->n_body = c->n_guarded = c->n_since_load = 0;
        {
            const long long frame_tiles = ((long long)c->H * c->W + R2L_TILE_RAYS - 1) / R2L_TILE_RAYS;
            int rc = ensure_x(c, (int)(frame_tiles < R2L_SLICE_TILES ? frame_tiles : R2L_SLICE_TILES), false);
            if (rc) return rc;
        }
    } else {
        pack_image_host(c, mode, img);
    }
    if (c->d_img[mode]) {
        (void)hipFree(c->d_img[mode]);
        c->d_img[mode] = nullptr;
    }
    hipError_t e = hipMalloc((void**)&c->d_img[mode], img.size());
    if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "hipMalloc image: %s", hipGetErrorString(e));
    e = hipMemcpy(c->d_img[mode], img.data(), img.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "hipMemcpy image: %s", hipGetErrorString(e));
    c->img_bytes[mode] = img.size();
    return R2L_OK;
}

// host-only: the packed chunk stream of r2l_common.h (no GPU needed; also behind r2l_debug_pack_host)
static void pack_image_host(const r2l_ctx* c, int mode, std::vector<char>& img) {
    const int np = np_of(mode);
    const int CH = r2l_chunk_bytes(np);
    const int AUX = R2L_FRAGS * np * R2L_FRAG_BYTES;
    const int cpt = r2l_chunks_per_tile(c->n_block);
    img.assign((size_t)cpt * CH, 0);
    auto aux = [&](int chunk) { return reinterpret_cast<float*>(img.data() + (size_t)chunk * CH + AUX); };
    const float Sa = c->act_scale;
    // head: chunk = k-step s, fragment = row tile u
    {
        const float* Wh = c->host_w[0].data();
        const float* bh = c->host_w[1].data();
        const float Sw = r2l_pow2_scale(Wh, (size_t)R2L_WIDTH * R2L_IN);
        const float S = Sa * Sw;
        for (int ch = 0; ch < R2L_HEAD_CHUNKS; ++ch) {
            char* chunk = img.data() + (size_t)ch * CH;
            for (int u = 0; u < R2L_RTILES; ++u)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        int col = r2l_head_col(ch, lane >> 4, j);
                        float v = col < 0 ? 0.f : Wh[(size_t)(16 * u + (lane & 15)) * R2L_IN + col] * Sw;
                        put_frag(chunk, np, u, lane, j, v);
                    }
        }
        for (int n = 0; n < R2L_WIDTH; ++n) aux(0)[n] = bh[n] * S;
        aux(R2L_HEAD_CHUNKS - 1)[32] = 1.0f / S;
    }
    // body: chunk m of a layer = row tiles 2m, 2m+1; fragment = (u&1)*8 + k-step
    for (int li = 0; li < 2 * c->n_block; ++li) {
        const float* Wl = c->host_w[2 + 2 * li].data();
        const float* bl = c->host_w[3 + 2 * li].data();
        const float Sw = r2l_pow2_scale(Wl, (size_t)R2L_WIDTH * R2L_WIDTH);
        const float S = Sa * Sw;
        for (int m = 0; m < R2L_RTILES / 2; ++m) {
            const int ci = R2L_HEAD_CHUNKS + li * (R2L_RTILES / 2) + m;
            char* chunk = img.data() + (size_t)ci * CH;
            for (int f = 0; f < R2L_FRAGS; ++f) {
                const int u = 2 * m + (f >> 3), ks = f & 7;
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        int k = r2l_kappa(ks, lane >> 4, j);
                        put_frag(chunk, np, f, lane, j, Wl[(size_t)(16 * u + (lane & 15)) * R2L_WIDTH + k] * Sw);
                    }
            }
            for (int i = 0; i < 32; ++i) aux(ci)[i] = bl[32 * m + i] * S;
            aux(ci)[32] = 1.0f / S;
        }
    }
    // tail: row tile 0 (fragments 0..7), rows 0..2 real
    {
        const int ti = 2 + 4 * c->n_block;
        const float* Wt = c->host_w[ti].data();
        const float* bt = c->host_w[ti + 1].data();
        const float Sw = r2l_pow2_scale(Wt, (size_t)3 * R2L_WIDTH);
        const float S = Sa * Sw;
        const int ci = cpt - 1;
        char* chunk = img.data() + (size_t)ci * CH;
        for (int ks = 0; ks < R2L_KSTEPS; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    int r = lane & 15;
                    int k = r2l_kappa(ks, lane >> 4, j);
                    put_frag(chunk, np, ks, lane, j, r < 3 ? Wt[(size_t)r * R2L_WIDTH + k] * Sw : 0.f);
                }
        for (int i = 0; i < 3; ++i) aux(ci)[i] = bt[i] * S;
        aux(ci)[32] = 1.0f / S;
    }
}

// ---- FP16_FP8 body stream (gen/body_gen.py pack_body_image is the Python restatement) -------------------
// Per ResMLP block: 16 chunks of 28 KiB (layer 1: 8, layer 2: 8); chunk u = row tile u (32 output features); 28 pieces of
// 1 KiB: piece s = fp16 hi fragment of k-step s (UNSCALED weights, 64 lanes x 16 B: lane = 32 h + row, elements
// r2l_kappa32(s, h, j)); piece 16 + j = the first 16 B/lane of bf6 (e3m2) operand j of the row tile, piece 24 + (j>>1)
// holds the last 8 B/lane of operands j (bytes 0..511) and j+1 (512..1023).  j = (term, t) in the order
// (0,0) (1,0) (0,1) (1,1) (0,2) (1,2) (0,3) (1,3): term 0 = (w - hi(w)) / 2^(e-16), term 1 = w / 2^(e-4), e = exponent of
// the layer's max|w|; 32 elements of 6 bits per lane, element i at bits [6i, 6i+6), element i = input feature
// r2l_mix32(t, h, i).  The E8M0 scales that undo the shifts travel in the aux block: 4 KiB per block =
// 256 f32 bias of layer 1 (act_scale domain, with the layer-2 biases of all earlier blocks folded in:
// x~_i = x_i - sum_{j<i} b2_j) | 4 x (swl1, sw1, swl2, sw2) | pad.
// Tail: [3,256] W_t / act_scale, then b_t + W_t sum_j b2_j.
#define R2L_BODY_CHUNK 28672
// Round-3 experiment "bf6r" (gen/body_gen.py configure; -DR2L_BF6R_STREAM): the term-1 operands bf6(W) are neither streamed
// nor read from LDS -- every wave converts them from the fp16 fragments it holds for its own MFMAs
// (v_cvt_scalef32_pk32_bf6_f16).  Chunk = 22 KiB: 16 fp16 fragments, then the four term-0 operands (w - hi(w)) / 2^(e-16) back
// to back, 1,536 B each: 1 KiB of (lane x 16 B), then 512 B of (lane x 8 B).  Parity green on hardware, 5 % slower than the
// 28 KiB stream above, which therefore stays the shipping form (profiles/r03_dma_experiments.txt).
#define R2L_BODYW_CHUNK 22528
// R2L_PREC_FP16_E4M3: the same stream with both operands of the correction terms in OCP e4m3: chunks of 32 KiB = 32 pieces;
// operand j: piece 16 + 2 j = bytes 0..15 of every lane, piece 17 + 2 j = bytes 16..31; byte i = element i;
// term 0 = (w - hi(w)) / 2^(e-20), term 1 = w / 2^(e-8) (e4m3 holds 448)
#define R2L_BODY8_CHUNK 32768
int r2l_layer_exponent(const float* w, size_t n) {
    float m = 0.f;
    for (size_t i = 0; i < n; ++i) {
        const float a = fabsf(w[i]);
        if (a > m && isfinite(a)) m = a;
    }
    if (m == 0.f) return -4;
    int e;
    frexpf(m, &e);
    return e;
}

// OCP bf6 = e3m2 (bias 3, max 28, subnormal step 2^-4), round to nearest even, saturating
unsigned r2l_f_to_bf6(double v) {
    const unsigned sgn = signbit(v) ? 32u : 0u;
    const double a = fabs(v);
    if (!(a == a)) return sgn | 31u;
    if (a >= 28.0) return sgn | 31u;
    int e;
    frexp(a, &e);
    int E = e - 1;                       // a = 1.x * 2^E
    if (a == 0.0 || E < -2) return sgn | (unsigned)nearbyint(ldexp(a, 4));   // subnormal: m/4 * 2^-2 (4 -> min normal)
    int qn = (int)nearbyint(ldexp(a, 2 - E));  // 4..8
    if (qn == 8) {
        qn = 4;
        ++E;
    }
    if (E > 4) return sgn | 31u;
    return sgn | (unsigned)(((E + 3) << 2) | (qn - 4));
}

// FP16_FP8 head stream of r2l_head_kernel (layout: r2l_common.h r2l_head_col32; restated in gen/head_gen.py pack_head)
static int pack_head_v1(const r2l_ctx* c, std::vector<char>& out, int f16) {
    const size_t CH = f16 ? R2L_HEADX_CHUNK : 28672, stream_bytes = 32 * CH;
    out.assign(stream_bytes + R2L_HEAD_AUX_BYTES, 0);
    const float* Wh = c->host_w[0].data();   // [256, 1008]
    const float* bh = c->host_w[1].data();
    const float Sa = c->act_scale;
    std::vector<float> Ws((size_t)R2L_WIDTH * R2L_IN);
    for (size_t i = 0; i < Ws.size(); ++i) Ws[i] = Wh[i] * Sa;   // exact: act_scale is a power of two
    const int e = r2l_layer_exponent(Ws.data(), Ws.size());
    if (e < -12 || e > 6)
        return r2l_set_error(R2L_EINVAL, "head layer: max|w| x act_scale = 2^%d is outside the range the fp16 + bf6 weight "
                             "split covers (2^-12 .. 2^6); use R2L_PREC_FP16X3", e);
    const int el = e - 16, ew = e - 4;
    uint32_t* aux = reinterpret_cast<uint32_t*>(out.data() + stream_bytes);
    for (int n = 0; n < 256; ++n) {
        const float v = (float)((double)bh[n] * Sa);
        memcpy(&aux[n], &v, 4);
    }
    for (int q = 0; q < 4; ++q) {
        aux[256 + 4 * q] = 0x01010101u * (uint32_t)(127 + el);
        aux[256 + 4 * q + 1] = 0x01010101u * (uint32_t)(127 + ew);
    }
    for (int p = 0; p < 16; ++p)
        for (int u = 0; u < 8; ++u) {
            char* chunk = out.data() + (size_t)(2 * p + (u >> 2)) * CH;
            const int k = u & 3;
            for (int lane = 0; lane < 64; ++lane) {
                const int h = lane >> 5;
                const float* row = Ws.data() + (size_t)(32 * u + (lane & 31)) * R2L_IN;
                uint64_t bits[2][3] = {{0, 0, 0}, {0, 0, 0}};
                for (int s = 0; s < 4; ++s) {
                    _Float16* ph = reinterpret_cast<_Float16*>(chunk + (size_t)(4 * k + s) * 1024 + lane * 16);
                    for (int j = 0; j < 8; ++j) {
                        const int col = r2l_head_col32(p, s, h, j);
                        const float w = col < 0 ? 0.f : row[col];
                        const _Float16 hi = (_Float16)w;
                        ph[j] = hi;
                        if (f16) {      // FP16X3_ASM: the residual as a second fp16 fragment
                            reinterpret_cast<_Float16*>(chunk + (size_t)(16 + 4 * k + s) * 1024 + lane * 16)[j] =
                                (_Float16)((double)w - (double)(float)hi);
                            continue;
                        }
                        const int i = 8 * s + j, bit = 6 * i, wd = bit >> 6, sh = bit & 63;
                        const uint64_t code[2] = {r2l_f_to_bf6(ldexp((double)w - (double)(float)hi, -el)),
                                                  r2l_f_to_bf6(ldexp((double)w, -ew))};
                        for (int t = 0; t < 2; ++t) {
                            bits[t][wd] |= code[t] << sh;
                            if (sh > 58) bits[t][wd + 1] |= code[t] >> (64 - sh);
                        }
                    }
                }
                for (int t = 0; t < 2 && !f16; ++t) {
                    memcpy(chunk + (size_t)(16 + 2 * k + t) * 1024 + lane * 16, bits[t], 16);
                    memcpy(chunk + (size_t)(24 + k) * 1024 + t * 512 + lane * 8, &bits[t][2], 8);
                }
            }
        }
    return R2L_OK;
}

static int pack_body_v3(const r2l_ctx* c, int fmt, std::vector<char>& out, size_t* aux_off, size_t* tail_off) {
    const int nb = c->n_block;
    const int e4m3 = fmt == R2L_STREAM_E4M3, wconv = fmt == R2L_STREAM_BF6R, f16 = fmt == R2L_STREAM_F16;
    const size_t CH = (e4m3 || f16) ? R2L_BODY8_CHUNK : (wconv ? R2L_BODYW_CHUNK : R2L_BODY_CHUNK), AUXB = 4096;
    const size_t stream = (size_t)nb * 16 * CH;
    *aux_off = stream;
    *tail_off = stream + (size_t)nb * AUXB;
    out.assign(*tail_off + 4096, 0);   // the body kernel's fused tail copies 4 KiB of it to LDS
    std::vector<double> Bsum(256, 0.0);
    const float Sa = c->act_scale;
    // FP16X3_ASM: the stream holds W x 2^8 (exact), so that the residuals of weights ~ 2^-5 are normal fp16 numbers; the
    // accumulators (hence the bias) carry the factor, the kernel's epilogue takes it out (gen/body_gen.py F16_WSHIFT)
    const float wsc = f16 ? 256.0f : 1.0f;
    for (int b = 0; b < nb; ++b) {
        const float* W[2] = {c->host_w[2 + 4 * b].data(), c->host_w[4 + 4 * b].data()};
        const float* b1 = c->host_w[3 + 4 * b].data();
        const float* b2 = c->host_w[5 + 4 * b].data();
        uint32_t* aux = reinterpret_cast<uint32_t*>(out.data() + *aux_off + (size_t)b * AUXB);
        for (int n = 0; n < 256; ++n) {
            double acc = b1[n];
            for (int k = 0; k < 256; ++k) acc += (double)W[0][(size_t)n * 256 + k] * Bsum[k];
            const float v = (float)(acc * Sa * wsc);
            memcpy(&aux[n], &v, 4);
        }
        for (int half = 0; half < 2; ++half)      // activation exponents: IN set, H set, next IN set (the last block's: block 0's)
            for (int i = 0; i < 3; ++i) {
                const int j = (b == nb - 1 && i == 2) ? 0 : 2 * b + i;
                aux[R2L_BODY_AUX_ACT / 4 + 4 * half + i] = (uint32_t)(127 + (j < (int)c->act.size() ? c->act[j] : R2L_ACT_EXP));
            }
        for (int layer = 0; layer < 2; ++layer) {
            const float* Wl = W[layer];
            const int e = r2l_layer_exponent(Wl, 65536);
            if (e < -12 || e > 6)
                return r2l_set_error(R2L_EINVAL, "body block %d layer %d: max|w| = 2^%d is outside the range the fp16 + bf6 "
                                     "weight split covers (2^-12 .. 2^6); use R2L_PREC_FP16X3", b, layer, e);
            const int el = e4m3 ? e - 20 : e - 16, ew = e4m3 ? e - 8 : e - 4;
            for (int q = 0; q < 4; ++q) {
                aux[256 + 4 * q + 2 * layer] = 0x01010101u * (uint32_t)(127 + el);
                aux[256 + 4 * q + 2 * layer + 1] = 0x01010101u * (uint32_t)(127 + ew);
            }
            for (int u = 0; u < 8; ++u) {
                char* chunk = out.data() + ((size_t)(b * 2 + layer) * 8 + u) * CH;
                for (int lane = 0; lane < 64; ++lane) {
                    const int h = lane >> 5;
                    const float* row = Wl + (size_t)(32 * u + (lane & 31)) * 256;
                    for (int s = 0; s < 16; ++s) {
                        _Float16* ph = reinterpret_cast<_Float16*>(chunk + (size_t)s * 1024 + lane * 16);
                        for (int j = 0; j < 8; ++j) ph[j] = (_Float16)(row[r2l_kappa32(s, h, j)] * wsc);
                    }
                    for (int s = 0; f16 && s < 16; ++s) {      // FP16X3_ASM: pieces 16 + s = the fp16 residuals w - hi(w)
                        _Float16* pl = reinterpret_cast<_Float16*>(chunk + (size_t)(16 + s) * 1024 + lane * 16);
                        for (int j = 0; j < 8; ++j) {
                            const float w = row[r2l_kappa32(s, h, j)] * wsc;
                            pl[j] = (_Float16)((double)w - (double)(float)(_Float16)w);
                        }
                    }
                    for (int j = 0; e4m3 && j < 8; ++j) {
                        const int term = j & 1, t = j >> 1;
                        unsigned char codes[32];
                        for (int el_i = 0; el_i < 32; ++el_i) {
                            const float w = row[r2l_mix32(t, h, el_i)];
                            const float hi = (float)(_Float16)w;
                            codes[el_i] = r2l_f_to_e4m3(term == 0 ? ldexp((double)w - (double)hi, -el) : ldexp((double)w, -ew));
                        }
                        memcpy(chunk + (size_t)(16 + 2 * j) * 1024 + lane * 16, codes, 16);
                        memcpy(chunk + (size_t)(17 + 2 * j) * 1024 + lane * 16, codes + 16, 16);
                    }
                    for (int j = 0; !e4m3 && !f16 && j < 8; ++j) {
                        const int term = j & 1, t = j >> 1;
                        uint64_t bits[3] = {0, 0, 0};
                        for (int el_i = 0; el_i < 32; ++el_i) {
                            const float w = row[r2l_mix32(t, h, el_i)];
                            const float hi = (float)(_Float16)w;
                            const double v = term == 0 ? ldexp((double)w - (double)hi, -el) : ldexp((double)w, -ew);
                            const uint64_t code = r2l_f_to_bf6(v);
                            const int bit = 6 * el_i, wd = bit >> 6, sh = bit & 63;
                            bits[wd] |= code << sh;
                            if (sh > 58) bits[wd + 1] |= code >> (64 - sh);
                        }
                        if (wconv) {
                            if (term == 1) continue;          // made on chip
                            char* op = chunk + 16384 + (size_t)t * 1536;
                            memcpy(op + lane * 16, bits, 16);
                            memcpy(op + 1024 + lane * 8, &bits[2], 8);
                        } else {
                            memcpy(chunk + (size_t)(16 + j) * 1024 + lane * 16, bits, 16);
                            memcpy(chunk + (size_t)(24 + (j >> 1)) * 1024 + (j & 1) * 512 + lane * 8, &bits[2], 8);
                        }
                    }
                }
            }
        }
        for (int k = 0; k < 256; ++k) Bsum[k] += b2[k];
    }
    const int ti = 2 + 4 * nb;
    const float* Wt = c->host_w[ti].data();
    const float* bt = c->host_w[ti + 1].data();
    float* tw = reinterpret_cast<float*>(out.data() + *tail_off);
    for (int r = 0; r < 3; ++r) {
        double acc = bt[r];
        for (int k = 0; k < 256; ++k) {
            tw[r * 256 + k] = Wt[(size_t)r * 256 + k] / Sa;
            acc += (double)Wt[(size_t)r * 256 + k] * Bsum[k];
        }
        tw[768 + r] = (float)acc;
        tw[772 + r] = (float)acc;   // second copy: the LDS reads of lanes 32..63 are 16 bytes further on
    }
    return R2L_OK;
}

int r2l_load_weights(r2l_ctx* c, const float* const* tensors, int n_tensors) {
    if (!c || !tensors) return r2l_set_error(R2L_EINVAL, "NULL argument");
    const int expect = 4 + 4 * c->n_block;
    if (n_tensors != expect)
        return r2l_set_error(R2L_EINVAL, "expected %d tensors (head, %d ResMLP blocks x 4, tail), got %d", expect,
                             c->n_block, n_tensors);
    c->host_w.clear();
    for (int i = 0; i < n_tensors; ++i) {
        size_t n;
        if (i == 0) n = (size_t)R2L_WIDTH * R2L_IN;
        else if (i == 1) n = R2L_WIDTH;
        else if (i == expect - 2) n = 3 * R2L_WIDTH;
        else if (i == expect - 1) n = 3;
        else n = ((i - 2) % 2 == 0) ? (size_t)R2L_WIDTH * R2L_WIDTH : R2L_WIDTH;
        if (!tensors[i]) return r2l_set_error(R2L_EINVAL, "tensor %d is NULL", i);
        c->host_w.emplace_back(tensors[i], tensors[i] + n);
    }
    for (int m = 0; m < R2L_N_MODES; ++m)
        if (c->d_img[m]) {
            (void)hipFree(c->d_img[m]);
            c->d_img[m] = nullptr;
        }
    c->act.assign((size_t)2 * c->n_block + 1, R2L_ACT_EXP);
    c->calib_pending = 1;   // the first FP16_FP8 render measures the activation ranges of these weights
    c->calib_started = 0;
    c->calib_calls = 0;
    int rc = build_image(c, c->mode);
    if (rc) return rc;
    c->loaded = true;
    return R2L_OK;
}

int r2l_set_act_exponents(r2l_ctx* c, const int* exps, int n) {
    if (!c) return r2l_set_error(R2L_EINVAL, "NULL ctx");
    if (!c->loaded) return r2l_set_error(R2L_ESTATE, "r2l_set_act_exponents before r2l_load_weights");
    if (!exps) {            // back to self-calibration on the next render
        c->calib_pending = 1;
        c->calib_started = 0;
        c->calib_calls = 0;
        return R2L_OK;
    }
    if (c->last_stream) (void)hipStreamSynchronize(c->last_stream);   // renders in flight read the aux blocks
    if (n != 2 * c->n_block + 1) return r2l_set_error(R2L_EINVAL, "expected %d exponents, got %d", 2 * c->n_block + 1, n);
    for (int i = 0; i < n; ++i)
        if (exps[i] < R2L_ACT_EXP_MIN || exps[i] > R2L_ACT_EXP_MAX)
            return r2l_set_error(R2L_EINVAL, "exponent %d = %d outside [%d, %d]", i, exps[i], R2L_ACT_EXP_MIN, R2L_ACT_EXP_MAX);
    c->act.assign(exps, exps + n);
    c->calib_pending = 0;
    if (c->d_body && c->n_block > 0) {   // one copy into the contiguous array, one launch that spreads it into the aux blocks
        std::vector<int> ex(c->act);
        ex[n - 1] = ex[0];
        hipError_t e = hipMemcpy(c->d_exps, ex.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = r2l_launch_spread_exponents(c->d_exps, c->n_block, c->d_body + c->aux_off, c->last_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->last_stream);
        if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "exponents to the device: %s", hipGetErrorString(e));
    }
    return R2L_OK;
}

int r2l_get_act_exponents(r2l_ctx* c, int* out, int n) {
    if (!c || !out) return r2l_set_error(R2L_EINVAL, "NULL argument");
    if (n != 2 * c->n_block + 1) return r2l_set_error(R2L_EINVAL, "expected room for %d exponents, got %d", 2 * c->n_block + 1, n);
    if (!c->d_body) {
        for (int i = 0; i < n; ++i) out[i] = i < (int)c->act.size() ? c->act[i] : R2L_ACT_EXP;
        return R2L_OK;
    }
    if (c->last_stream) (void)hipStreamSynchronize(c->last_stream);   // a calibration enqueued there writes them
    // what the kernel reads: after a device-side calibration it exists only there (the aux blocks and this array beside them)
    hipError_t e = hipMemcpy(out, c->d_exps, (size_t)n * sizeof(int), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "hipMemcpy exponents: %s", hipGetErrorString(e));
    out[2 * c->n_block] = out[0];   // the tail reads x itself; the slot holds the next tile's first input set
    return R2L_OK;
}

// ---- range tracking ------------------------------------------------------------------------------------
int r2l_set_guard_period(r2l_ctx* c, int period) {
    if (!c) return r2l_set_error(R2L_EINVAL, "NULL ctx");
    if (period < 0) return r2l_set_error(R2L_EINVAL, "guard period %d < 0", period);
    if (period > 0 && c->n_block > r2l_body_guard_max_blocks(1))
        return r2l_set_error(R2L_EINVAL, "the range guard keeps 2 n_block rows of maxima in LDS: n_block <= %d", r2l_body_guard_max_blocks(1));
    c->guard_period = period;
    c->n_since_load = 0;       // the next launch is the first of the new phase: guarded when period > 0
    return R2L_OK;
}

int r2l_get_range_status(r2l_ctx* c, r2l_range_status* out, int reset) {
    if (!c || !out) return r2l_set_error(R2L_EINVAL, "NULL argument");
    memset(out, 0, sizeof *out);
    out->worst_set = -1;
    if (!c->loaded || !c->d_range || !c->d_gstats || !scaled_mode(c->mode))
        return r2l_set_error(R2L_ESTATE, "r2l_get_range_status needs loaded weights and R2L_PREC_FP16_FP8 / R2L_PREC_FP16_E4M3 "
                                         "(the other modes have no operand scales to watch)");
    // ONE copy: range words | maxima of the guarded launches | exponents in use (one allocation, see r2l_ctx)
    const int nset = 2 * c->n_block;
    std::vector<unsigned> words((size_t)4 + 2 * (nset + 1));
    if (c->last_stream) (void)hipStreamSynchronize(c->last_stream);
    hipError_t e = hipMemcpy(words.data(), c->d_range, words.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "hipMemcpy range words: %s", hipGetErrorString(e));
    const unsigned* r = words.data();
    const unsigned* g = r + 4;
    const int* ex = reinterpret_cast<const int*>(g + nset + 1);
    auto f = [](unsigned u) { float v; memcpy(&v, &u, 4); return v; };
    // largest magnitude of the operand format: bf6 (e3m2) 28, e4m3 448; the calibration aims at <= 16 in both
    const float top = e4m3_terms(c->mode) ? 448.0f : 28.0f;
    out->h0_max = f(r[0]) / c->act_scale;
    out->h0_fill = c->n_block > 0 ? ldexpf(f(r[0]), -ex[0]) / top : 0.0f;
    out->stream_max = out->h0_max;
    for (int j = 0; j < nset; ++j) {
        if (f(g[j]) / c->act_scale > out->stream_max) out->stream_max = f(g[j]) / c->act_scale;
        const float fill = ldexpf(f(g[j]), -ex[j]) / top;
        if (fill > out->worst_fill) {
            out->worst_fill = fill;
            out->worst_set = j;
        }
    }
    const float m = out->h0_fill > out->worst_fill ? out->h0_fill : out->worst_fill;
    out->saturated = m >= 1.0f;
    out->beyond_calibration = m > 16.0f / top;
    out->format_top = top;
    out->launches = c->n_body;
    out->guarded_launches = c->n_guarded;
    if (reset) {
        e = hipMemset(c->d_range, 0, (size_t)(4 + nset + 1) * sizeof(unsigned));
        if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "hipMemset range words: %s", hipGetErrorString(e));
        c->n_body = c->n_guarded = 0;
    }
    return R2L_OK;
}

int r2l_recalibrate(r2l_ctx* c, void* stream) {
    if (!c) return r2l_set_error(R2L_EINVAL, "NULL ctx");
    if (!c->loaded || !c->d_gstats || c->n_block < 1)
        return r2l_set_error(R2L_ESTATE, "r2l_recalibrate needs loaded weights, R2L_PREC_FP16_FP8 and n_block >= 1");
    if (two_part(c->mode))      // its guarded launches see the blocks behind the split only
        return r2l_set_error(R2L_ESTATE, "r2l_recalibrate: R2L_PREC_FP16_SPLIT / _SPLIT8 borrow the exponents of R2L_PREC_FP16_FP8 -- recalibrate there");
    if (c->n_guarded < 1) return r2l_set_error(R2L_ESTATE, "r2l_recalibrate: no guarded launch since the last reset of the range words");
    hipError_t e = r2l_launch_recalibrate(c->d_gstats, c->d_range, c->n_block, c->d_body + c->aux_off, c->d_exps, (hipStream_t)stream);
    if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "recalibration launch: %s", hipGetErrorString(e));
    c->calib_pending = 0;
    c->last_stream = (hipStream_t)stream;
    return R2L_OK;
}

int r2l_set_split_block(r2l_ctx* c, int split_block) {
    if (!c) return r2l_set_error(R2L_EINVAL, "NULL ctx");
    if (split_block < 0 || split_block > c->n_block)
        return r2l_set_error(R2L_EINVAL, "split block %d outside [0, %d]", split_block, c->n_block);
    c->split_block = split_block;
    return R2L_OK;
}

int r2l_debug_set_fused_tail(r2l_ctx* c, int on) {
    if (!c) return r2l_set_error(R2L_EINVAL, "NULL ctx");
    c->fuse_tail = on ? 1 : 0;
    return R2L_OK;
}

// Host-only packing for tests (no GPU): tensors in state_dict order -> chunk stream bytes.
// Returns the byte count (or a negative code); copies min(count, cap) bytes into out.
long long r2l_debug_pack_host(const float* const* tensors, int n_tensors, int n_block, int precision_mode, char* out,
                              long long cap) {
    if (!tensors || n_tensors != 4 + 4 * n_block || n_block < 0) return r2l_set_error(R2L_EINVAL, "bad tensor list");
    if (!mode_ok(precision_mode)) return r2l_set_error(R2L_EINVAL, "bad mode");
    r2l_ctx c;
    c.n_block = n_block;
    c.act_scale = 16.0f;
    for (int i = 0; i < n_tensors; ++i) {
        size_t n;
        if (i == 0) n = (size_t)R2L_WIDTH * R2L_IN;
        else if (i == 1) n = R2L_WIDTH;
        else if (i == n_tensors - 2) n = 3 * R2L_WIDTH;
        else if (i == n_tensors - 1) n = 3;
        else n = ((i - 2) % 2 == 0) ? (size_t)R2L_WIDTH * R2L_WIDTH : R2L_WIDTH;
        c.host_w.emplace_back(tensors[i], tensors[i] + n);
    }
    std::vector<char> img;
    if (split_mode(precision_mode)) {   // the image of this mode's head launch (r2l_head_kernel)
        int rc = pack_head_v1(&c, img, head_x3(precision_mode));
        if (rc) return rc;
    } else {
        pack_image_host(&c, precision_mode, img);
    }
    if (out && cap > 0) memcpy(out, img.data(), (size_t)(cap < (long long)img.size() ? cap : (long long)img.size()));
    return (long long)img.size();
}

static thread_local int g_debug_pack_e4m3 = 0;
int r2l_debug_pack_body_format(int fmt) {   // which stream r2l_debug_pack_body_host packs: R2L_STREAM_*
    if (fmt < 0 || fmt > R2L_STREAM_F16) return r2l_set_error(R2L_EINVAL, "stream format %d", fmt);
    g_debug_pack_e4m3 = fmt;
    return R2L_OK;
}

// Host-only: the FP16_FP8 body stream (chunks | aux blocks | tail) of pack_body_v3, for the CPU tests that run
// gen/body_gen.py's emulator on exactly the bytes the GPU streams.  offs[0] = aux offset, offs[1] = tail offset.
long long r2l_debug_pack_body_host(const float* const* tensors, int n_tensors, int n_block, char* out, long long cap,
                                   long long* offs) {
    if (!tensors || n_tensors != 4 + 4 * n_block || n_block < 0) return r2l_set_error(R2L_EINVAL, "bad tensor list");
    r2l_ctx c;
    c.n_block = n_block;
    c.act_scale = 16.0f;
    for (int i = 0; i < n_tensors; ++i) {
        size_t n;
        if (i == 0) n = (size_t)R2L_WIDTH * R2L_IN;
        else if (i == 1) n = R2L_WIDTH;
        else if (i == n_tensors - 2) n = 3 * R2L_WIDTH;
        else if (i == n_tensors - 1) n = 3;
        else n = ((i - 2) % 2 == 0) ? (size_t)R2L_WIDTH * R2L_WIDTH : R2L_WIDTH;
        c.host_w.emplace_back(tensors[i], tensors[i] + n);
    }
    std::vector<char> img;
    size_t aux_off = 0, tail_off = 0;
    int rc = pack_body_v3(&c, g_debug_pack_e4m3, img, &aux_off, &tail_off);
    if (rc) return rc;
    if (offs) {
        offs[0] = (long long)aux_off;
        offs[1] = (long long)tail_off;
    }
    if (out && cap > 0) memcpy(out, img.data(), (size_t)(cap < (long long)img.size() ? cap : (long long)img.size()));
    return (long long)img.size();
}

// The hand-scheduled body alone: x_out = body(x_in) on n_tiles ray tiles in the register-image layout
// [tile][wave 4][u*2+c 32][lane 64][4] f32 (act_scale domain, layer-2 biases folded: see pack_body_v3).
int r2l_debug_body(r2l_ctx* c, const float* x_in_dev, float* x_out_dev, int n_tiles, void* stream) {
    if (!c || !x_in_dev || !x_out_dev || n_tiles < 1) return r2l_set_error(R2L_EINVAL, "bad argument to r2l_debug_body");
    if (!c->loaded || !split_mode(c->mode) || c->n_block < 1)
        return r2l_set_error(R2L_ESTATE, "r2l_debug_body needs loaded weights, R2L_PREC_FP16_FP8 / _E4M3 and n_block >= 1");
    if (c->calib_pending && scaled_mode(c->mode)) {   // as a render would: the exponents of these weights on this input
        c->calib_pending = 0;
        hipError_t ec = r2l_launch_calib(x_in_dev, c->d_wcal, c->n_block, n_tiles, c->act_scale, c->d_stats,
                                         c->d_body + c->aux_off, c->d_exps, 0, (hipStream_t)stream);
        if (ec != hipSuccess) return r2l_set_error(R2L_EHIP, "r2l calibration launch: %s", hipGetErrorString(ec));
    }
    R2LBodyParams pb;
    pb.wimg = c->d_body;
    pb.aux = c->d_body + c->aux_off;
    pb.xin = x_in_dev;
    pb.xout = x_out_dev;
    pb.n_tiles = n_tiles;
    pb.n_block = c->n_block;
    pb.rgb = nullptr;
    pb.tail = reinterpret_cast<const float*>(c->d_body + c->tail_off);
    pb.n_rays = 0;
    pb.tile_begin = 0;
    pb.e4m3 = e4m3_terms(c->mode) ? 1 : c->mode == R2L_PREC_FP16X3_ASM ? 2 : 0;
    pb.gstats = c->guard_period == 1 && scaled_mode(c->mode) && c->n_block <= r2l_body_guard_max_blocks(pb.e4m3) ? c->d_gstats : nullptr;
    c->last_stream = (hipStream_t)stream;
    hipError_t e = r2l_launch_body(pb, balanced_grid(n_tiles, c->n_cu), (hipStream_t)stream);
    if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "r2l body launch: %s", hipGetErrorString(e));
    return R2L_OK;
}

int r2l_set_precision(r2l_ctx* c, int mode) {
    if (!c) return r2l_set_error(R2L_EINVAL, "NULL ctx");
    if (!mode_ok(mode)) return r2l_set_error(R2L_EINVAL, "bad precision_mode %d", mode);
    if (split_mode(mode) && !default_acts(c))     // also when an image of that mode was packed before the slopes were set
        return r2l_set_error(R2L_EINVAL, "activation slopes head %g / inner %g / out %g: the generated kernels (fp16_fp8, fp16_e4m3, "
                                         "fp16x3_asm) are built for relu / relu / none; fp16x3 renders the others",
                             c->act_head, c->act_in, c->act_out);
    if (c->loaded && split_mode(mode) && scaled_mode(c->body_mode) && c->d_body && c->body_mode != mode && c->n_block > 0 &&
        !c->calib_pending) {
        // the other split mode's body stream is about to replace this one: the exponents the device calibrated live only
        // in its aux blocks -- carry them over (same operand sets, same meaning)
        std::vector<int> ex((size_t)2 * c->n_block + 1);
        int rc = r2l_get_act_exponents(c, ex.data(), (int)ex.size());
        if (rc) return rc;
        c->act = ex;
    }
    const int prev = c->mode;
    c->mode = mode;
    if (c->loaded && (!c->d_img[mode] || (split_mode(mode) && c->body_mode != mode))) {
        const int rc = build_image(c, mode);
        if (rc) c->mode = prev;      // weights this mode cannot pack: the context keeps rendering in the mode it had
        return rc;
    }
    return R2L_OK;
}

int r2l_set_activations(r2l_ctx* c, float head_slope, float inner_slope, float out_slope) {
    return r2l_set_network_form(c, head_slope, inner_slope, out_slope, 1);
}

int r2l_set_network_form(r2l_ctx* c, float head_slope, float inner_slope, float out_slope, int block_residual) {
    if (!c) return r2l_set_error(R2L_EINVAL, "NULL ctx");
    const float s[3] = {head_slope, inner_slope, out_slope};
    for (float v : s)
        if (!(v >= 0.0f && v <= 1.0f)) return r2l_set_error(R2L_EINVAL, "activation slope %g is outside [0, 1]", v);
    const bool dflt = head_slope == 0.0f && inner_slope == 0.0f && out_slope == 1.0f && block_residual;
    if (!dflt && split_mode(c->mode))
        return r2l_set_error(R2L_EINVAL, "activation slopes %g / %g / %g%s need a compiler-scheduled mode (R2L_PREC_FP16X3 / _FP16X1): the "
                                         "generated kernels of the current mode are built for relu / relu / none ResMLP blocks",
                             head_slope, inner_slope, out_slope, block_residual ? "" : " without the block residual");
    if (c->last_stream) (void)hipStreamSynchronize(c->last_stream);
    c->act_head = head_slope;
    c->act_in = inner_slope;
    c->act_out = out_slope;
    c->block_resid = block_residual ? 1 : 0;
    return R2L_OK;
}

int r2l_set_z_vals(r2l_ctx* c, const float* z_host, int n) {
    if (!c || !z_host) return r2l_set_error(R2L_EINVAL, "NULL argument");
    if (n != R2L_NSAMPLE) return r2l_set_error(R2L_EINVAL, "expected %d z values, got %d", R2L_NSAMPLE, n);
    memcpy(c->z, z_host, sizeof c->z);
    hipError_t e = hipMemcpy(c->d_z, c->z, sizeof c->z, hipMemcpyHostToDevice);
    if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "hipMemcpy z_vals: %s", hipGetErrorString(e));
    return R2L_OK;
}

// ---- launches --------------------------------------------------------------------------
static void fill_common(const r2l_ctx* c, R2LParams& p) {
    memset(&p, 0, sizeof p);
    p.z = c->d_z;
    p.focal = (float)c->focal;
    p.half_w = (float)(c->W * .5);
    p.half_h = (float)(c->H * .5);
    p.act_scale = c->act_scale;
    p.neg1 = -1.0f;
    p.W = c->W;
    p.n_block = c->n_block;
    p.use_residual = c->use_residual;
    p.act_head = c->act_head;
    p.act_in = c->act_in;
    p.act_out = c->act_out;
    p.block_resid = c->block_resid ? 1.0f : 0.0f;
    p.chunks_per_tile = r2l_chunks_per_tile(c->n_block);
    p.scratch = c->d_scratch;
}

static int timing_events(r2l_ctx* c, hipEvent_t* e0, hipEvent_t* e1) {
    if (c->ev_used + 2 > (int)c->ev.size()) {
        for (int i = 0; i < 2; ++i) {
            hipEvent_t e;
            hipError_t er = hipEventCreate(&e);
            if (er != hipSuccess) return r2l_set_error(R2L_EHIP, "hipEventCreate: %s", hipGetErrorString(er));
            c->ev.push_back(e);
        }
    }
    *e0 = c->ev[c->ev_used];
    *e1 = c->ev[c->ev_used + 1];
    c->ev_used += 2;
    return R2L_OK;
}

// FP16_FP8: head launch -> hand-scheduled body launch (which ends every ray tile with the tail layer when the network has
// the global skip; else -> tail launch) per slice of <= R2L_SLICE_TILES ray tiles; the slices reuse the library-owned h0
// buffer (1 KiB per ray), stream-ordered.  The timing events bracket the body launch (the dominant kernel).
static int ensure_x(r2l_ctx* c, int tiles, bool need_xb) {
    // d_xa is sized for the context's frame at r2l_load_weights; only a call with more rays than H x W (r2l_render_rays,
    // several poses) grows it -- hipFree synchronises the device, so such a call is not stream-ordered the first time.
    // d_xb exists only for the unfused form (networks without the global skip; r2l_debug_set_fused_tail(0)).
    auto grow = [&](float** buf, int* cap) -> int {
        if (tiles <= *cap) return R2L_OK;
        if (*buf) (void)hipFree(*buf);
        *buf = nullptr;
        *cap = 0;
        const size_t bytes = (size_t)tiles * R2L_TILE_RAYS * R2L_WIDTH * sizeof(float);
        hipError_t e = hipMalloc((void**)buf, bytes);
        if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "hipMalloc x buffer (%zu B): %s", bytes, hipGetErrorString(e));
        *cap = tiles;
        return R2L_OK;
    };
    int rc = grow(&c->d_xa, &c->x_tiles);
    if (rc == R2L_OK && need_xb) rc = grow(&c->d_xb, &c->xb_tiles);
    return rc;
}

static int launch_split(r2l_ctx* c, const R2LParams& p, hipStream_t s) {
    const int slice = p.n_tiles < R2L_SLICE_TILES ? p.n_tiles : R2L_SLICE_TILES;
    const bool fused_form = c->n_block > 0 && c->use_residual && c->fuse_tail;
    // R2L_PREC_FP16_SPLIT / _SPLIT8: blocks [0, split) on the three-pass kernel (x image out), blocks [split, n_block) on the bf6 / e4m3 kernel
    int split = -1;
    if (two_part(c->mode) && c->n_block > 0) {
        split = c->split_block < 0 ? c->n_block / 2 : c->split_block;
        if (split > c->n_block) split = c->n_block;
    }
    const bool two_bodies = split > 0 && split < c->n_block;
    int rc = ensure_x(c, slice, c->n_block > 0 && (!fused_form || two_bodies));
    if (rc) return rc;
    c->last_stream = s;
    for (int t0 = 0; t0 < p.n_tiles; t0 += R2L_SLICE_TILES) {
        const int nt = p.n_tiles - t0 < R2L_SLICE_TILES ? p.n_tiles - t0 : R2L_SLICE_TILES;
        const int grid = balanced_grid(nt, c->n_cu);
        R2LParams ph = p;
        ph.wimg = c->d_img[c->mode];
        ph.xbuf = c->d_xa;
        ph.tile_begin = t0;
        ph.n_tiles = nt;
        ph.range = c->d_range;     // every ray's h0 enters the running maximum (r2l_get_range_status)
        hipError_t e = r2l_launch_head(ph, grid, s, head_x3(c->mode));
        if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "r2l head launch: %s", hipGetErrorString(e));
        const float* body_out = c->d_xa;
        bool fused = false;
        if (c->n_block > 0 && c->calib_pending && scaled_mode(c->mode)) {
            // activation exponents from this call's own head output: device work in stream order, no host round trip
            // a call of fewer than R2L_CALIB_TILES ray tiles is a thin sample: its maxima count, but the measurement stays open
            // and the next call adds its own (exponents only grow), until one call has filled the sample
            e = r2l_launch_calib(c->d_xa, c->d_wcal, c->n_block, nt, c->act_scale, c->d_stats, c->d_body + c->aux_off,
                                 c->d_exps, c->calib_started, s);
            c->calib_started = 1;
            if (nt >= R2L_CALIB_TILES || ++c->calib_calls >= R2L_CALIB_MAX_CALLS) c->calib_pending = 0;
            if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "r2l calibration launch: %s", hipGetErrorString(e));
        }
        if (c->n_block > 0) {
            R2LBodyParams pb;
            pb.wimg = c->d_body;
            pb.aux = c->d_body + c->aux_off;
            pb.xin = c->d_xa;
            pb.n_tiles = nt;
            pb.n_block = c->n_block;
            // with the global skip the body kernel finishes the rays itself: its fused tail reads h back from the image operand
            // xout points to (nothing is written there in that form)
            fused = c->use_residual && c->fuse_tail;
            pb.rgb = fused ? p.rgb : nullptr;
            pb.xout = fused ? c->d_xa : c->d_xb;
            pb.tail = reinterpret_cast<const float*>(c->d_body + c->tail_off);
            pb.n_rays = p.n_rays;
            pb.tile_begin = t0;
            pb.e4m3 = e4m3_terms(c->mode) ? 1 : c->mode == R2L_PREC_FP16X3_ASM ? 2 : 0;
            R2LBodyParams p2 = pb;
            if (split >= 0) {
                // blocks [0, split) in three passes (stream d_body2), blocks [split, n_block) with bf6 / e4m3 terms (stream d_body)
                const int n3 = split, n6 = c->n_block - split;
                R2LBodyParams& q6 = two_bodies ? p2 : pb;
                if (two_bodies) {
                    pb.rgb = nullptr;
                    pb.xout = c->d_xb;
                    p2.xin = c->d_xb;                               // the second launch continues the stream in place
                }
                if (n3 > 0) {
                    pb.wimg = c->d_body2;
                    pb.aux = c->d_body2 + c->aux_off2;
                    pb.tail = reinterpret_cast<const float*>(c->d_body2 + c->tail_off2);
                    pb.e4m3 = 2;
                    pb.n_block = n3;
                }
                if (n6 > 0) {
                    q6.wimg = c->d_body + (size_t)split * 16 * (e4m3_terms(c->mode) ? R2L_BODY8_CHUNK : R2L_BF6_CHUNK);
                    q6.aux = c->d_body + c->aux_off;
                    q6.tail = reinterpret_cast<const float*>(c->d_body + c->tail_off);
                    q6.e4m3 = e4m3_terms(c->mode) ? 1 : 0;
                    q6.n_block = n6;
                    if (split > 0) {
                        // the kernel converts the NEXT ray tile's x with the exponent its last block's aux names (in the full
                        // stream: block 0's input set): a launch that starts at `split` needs that block's there -- a patched copy
                        // of its aux blocks, made in stream order behind whatever calibrated them
                        q6.aux = c->d_body2 + c->aux_split_off;
                        e = r2l_launch_split_aux(c->d_body + c->aux_off, split, n6, c->d_body2 + c->aux_split_off, s);
                        if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "r2l split aux launch: %s", hipGetErrorString(e));
                    }
                }
            }
            // range guard: the first launch after a weight load, then every guard_period-th (r2l_set_guard_period); of a split
            // launch pair the bf6 part, whose rows of maxima start at its first block's
            auto guarded = [&](R2LBodyParams& q, int first) {
                const bool guard = c->guard_period > 0 && scaled_mode(c->mode) && q.e4m3 != 2 &&
                                   c->n_block <= r2l_body_guard_max_blocks(q.e4m3) && c->n_since_load % c->guard_period == 0;
                q.gstats = guard ? c->d_gstats + 2 * first : nullptr;
                return guard;
            };
            const bool g1 = guarded(pb, 0), g2 = two_bodies && guarded(p2, split);
            const bool guard = g1 || g2;
            ++c->n_since_load;
            ++c->n_body;
            c->n_guarded += guard ? 1 : 0;
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (c->timing) {
                rc = timing_events(c, &e0, &e1);
                if (rc) return rc;
                (void)hipEventRecord(e0, s);
            }
            e = r2l_launch_body(pb, grid, s);
            if (e == hipSuccess && two_bodies) e = r2l_launch_body(p2, grid, s);
            if (c->timing) (void)hipEventRecord(e1, s);
            if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "r2l body launch: %s", hipGetErrorString(e));
            body_out = c->d_xb;
        }
        if (fused) continue;
        R2LTailParams pt;
        pt.xa = c->use_residual ? c->d_xa : nullptr;
        pt.xb = body_out;
        pt.wt = reinterpret_cast<const float*>(c->d_body + c->tail_off);
        pt.rgb = p.rgb;
        pt.n_tiles = nt;
        pt.tile_begin = t0;
        pt.n_rays = p.n_rays;
        e = r2l_launch_tail(pt, s);
        if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "r2l tail launch: %s", hipGetErrorString(e));
    }
    return R2L_OK;
}

static int timed_launch(r2l_ctx* c, const R2LParams& p, hipStream_t s) {
    if (split_mode(c->mode)) return launch_split(c, p, s);
    const int grid = balanced_grid(p.n_tiles, c->n_cu);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->timing) {
        int rc = timing_events(c, &e0, &e1);
        if (rc) return rc;
        (void)hipEventRecord(e0, s);
    }
    hipError_t e = r2l_launch_resmlp(p, c->mode, grid, s);
    if (c->timing) (void)hipEventRecord(e1, s);
    if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "r2l_resmlp launch: %s", hipGetErrorString(e));
    return R2L_OK;
}

int r2l_render(r2l_ctx* c, const float* c2w, int c2w_on_device, int n_pose, int row_begin, int row_end,
               float* rgb_out_dev, void* stream) {
    if (!c || !c2w || !rgb_out_dev) return r2l_set_error(R2L_EINVAL, "NULL argument");
    if (!c->loaded) return r2l_set_error(R2L_ESTATE, "r2l_render before r2l_load_weights");
    if (row_begin < 0 || row_end > c->H || row_begin >= row_end)
        return r2l_set_error(R2L_EINVAL, "bad row range [%d,%d) for H=%d", row_begin, row_end, c->H);
    if (n_pose < 1 || (!c2w_on_device && n_pose != 1))
        return r2l_set_error(R2L_EINVAL, "n_pose=%d (host poses are rendered one per call)", n_pose);
    const long long rpp = (long long)(row_end - row_begin) * c->W;
    const long long total = rpp * n_pose;
    if (total > 0x7fffffffLL) return r2l_set_error(R2L_EINVAL, "too many rays in one call: %lld", total);
    R2LParams p;
    fill_common(c, p);
    p.wimg = c->d_img[c->mode];
    p.rgb = rgb_out_dev;
    if (c2w_on_device) p.c2w = c2w;
    else memcpy(p.c2w_host, c2w, sizeof p.c2w_host);
    p.pix_begin = row_begin * c->W;
    p.rays_per_pose = (int)rpp;
    p.n_rays = (int)total;
    p.n_tiles = (int)((total + R2L_TILE_RAYS - 1) / R2L_TILE_RAYS);
    return timed_launch(c, p, (hipStream_t)stream);
}

int r2l_render_rays(r2l_ctx* c, const float* rays_o_dev, const float* rays_d_dev, int n, float* rgb_out_dev,
                    void* stream) {
    if (!c || !rays_o_dev || !rays_d_dev || !rgb_out_dev) return r2l_set_error(R2L_EINVAL, "NULL argument");
    if (!c->loaded) return r2l_set_error(R2L_ESTATE, "r2l_render_rays before r2l_load_weights");
    if (n < 0) return r2l_set_error(R2L_EINVAL, "n=%d", n);
    if (n == 0) return R2L_OK;
    R2LParams p;
    fill_common(c, p);
    p.wimg = c->d_img[c->mode];
    p.rgb = rgb_out_dev;
    p.rays_o = rays_o_dev;
    p.rays_d = rays_d_dev;
    p.rays_per_pose = n;
    p.n_rays = n;
    p.n_tiles = (n + R2L_TILE_RAYS - 1) / R2L_TILE_RAYS;
    return timed_launch(c, p, (hipStream_t)stream);
}

int r2l_sample_embed(r2l_ctx* c, const float* c2w_host, int row_begin, int row_end, float* pts_out_dev,
                     float* emb_out_dev, void* stream) {
    if (!c || !c2w_host) return r2l_set_error(R2L_EINVAL, "NULL argument");
    if (row_begin < 0 || row_end > c->H || row_begin >= row_end)
        return r2l_set_error(R2L_EINVAL, "bad row range [%d,%d) for H=%d", row_begin, row_end, c->H);
    R2LParams p;
    fill_common(c, p);
    memcpy(p.c2w_host, c2w_host, sizeof p.c2w_host);
    p.pix_begin = row_begin * c->W;
    p.rays_per_pose = (row_end - row_begin) * c->W;
    p.n_rays = p.rays_per_pose;
    hipError_t e = r2l_launch_sample_embed(p, pts_out_dev, emb_out_dev, (hipStream_t)stream);
    if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "sample_embed launch: %s", hipGetErrorString(e));
    return R2L_OK;
}

int r2l_embed(const float* x_dev, int n, int dim, int L, float* emb_out_dev, void* stream) {
    if (!x_dev || !emb_out_dev || n < 0 || dim <= 0 || L <= 0 || L > 16)
        return r2l_set_error(R2L_EINVAL, "bad argument to r2l_embed");
    int rc = r2l_require_gfx950(nullptr);
    if (rc) return rc;
    if (n == 0) return R2L_OK;
    hipError_t e = r2l_launch_embed(x_dev, (long long)n * dim, L, emb_out_dev, (hipStream_t)stream);
    if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "embed launch: %s", hipGetErrorString(e));
    return R2L_OK;
}

long long r2l_flops_per_ray(const r2l_ctx* c) {
    if (!c) return 0;
    return 2LL * ((long long)R2L_IN * R2L_WIDTH + 2LL * c->n_block * R2L_WIDTH * R2L_WIDTH + R2L_WIDTH * 3);
}
long long r2l_kernel_flops_per_ray(const r2l_ctx* c) {
    if (!c) return 0;
    if (split_mode(c->mode))   // r2l_body_kernel: the 2 n_block body layers, and the tail layer when it is fused in
        return 2LL * 2LL * c->n_block * R2L_WIDTH * R2L_WIDTH + (c->use_residual && c->fuse_tail && c->n_block > 0 ? 2LL * 3 * R2L_WIDTH : 0);
    return r2l_flops_per_ray(c);
}
long long r2l_weight_image_bytes(const r2l_ctx* c) {
    if (!c) return 0;
    return (long long)c->img_bytes[c->mode] + (split_mode(c->mode) ? (long long)c->body_bytes : 0) +
           (two_part(c->mode) ? (long long)c->body2_bytes : 0);
}
int r2l_rays_per_tile(const r2l_ctx*) { return R2L_TILE_RAYS; }

int r2l_timing_enable(r2l_ctx* c, int on) {
    if (!c) return r2l_set_error(R2L_EINVAL, "NULL ctx");
    c->timing = on != 0;
    return R2L_OK;
}

int r2l_kernel_time_ms(r2l_ctx* c, double* total_ms, int* n_launches, int reset) {
    if (!c) return r2l_set_error(R2L_EINVAL, "NULL ctx");
    double tot = 0;
    for (int i = 0; i + 1 < c->ev_used; i += 2) {
        hipError_t e = hipEventSynchronize(c->ev[i + 1]);
        if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "hipEventSynchronize: %s", hipGetErrorString(e));
        float ms = 0;
        e = hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]);
        if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "hipEventElapsedTime: %s", hipGetErrorString(e));
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (n_launches) *n_launches = c->ev_used / 2;
    if (reset) c->ev_used = 0;
    return R2L_OK;
}
