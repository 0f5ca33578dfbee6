// C-ABI host side for the NeRF teacher (include/r2l_hip.h): context, host-side packing of
// the 24 state_dict tensors into the MFMA fragment stream (nerf_common.h), and the
// coarse -> importance-sample -> fine pipeline of render_rays (main.py:624-756).
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include <functional>
#include <vector>

#include "../../include/r2l_hip.h"
#include "nerf_common.h"
#include "nerf_kernels.h"
#include "r2l_host_util.h"

namespace {

struct PackedNet {
    char* d_img[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // [precision mode]
    float inv_scale[8][NERF_N_SCALES];
    char* d_img_alpha = nullptr;             // the three-pass stream without the view branch (coarse network, nerf_set_skip_rgb0)
    char* d_img_exit = nullptr;              // the fine network's stream with the second exit behind the density (fp16x3_asm or fp16_mix), and
    int exit_mode = -1;                      // the mode it was packed for
    std::vector<std::vector<float>> host_w;  // 24 tensors, state_dict order
    bool loaded = false;
};

// state_dict order (model/nerf_raybased.py:357-375)
enum { T_PTS0_W = 0, T_VIEWS_W = 16, T_VIEWS_B = 17, T_FEAT_W = 18, T_FEAT_B = 19, T_ALPHA_W = 20, T_ALPHA_B = 21,
       T_RGB_W = 22, T_RGB_B = 23 };
const size_t kTensorNumel[24] = {
    256 * 63, 256, 256 * 256, 256, 256 * 256, 256, 256 * 256, 256, 256 * 256, 256,  // pts_linears 0..4
    256 * 319, 256, 256 * 256, 256, 256 * 256, 256,                                    // pts_linears 5..7
    128 * 283, 128, 256 * 256, 256, 256, 1, 3 * 128, 3};

int np_of(int mode) { return mode == R2L_PREC_FP16X1 ? 1 : 2; }
bool mode_ok(int mode) {
    return mode == R2L_PREC_FP16X3 || mode == R2L_PREC_FP16X1 || mode == R2L_PREC_FP16_FP8 || mode == R2L_PREC_FP16X3_ASM || mode == R2L_PREC_FP16_MIX;
}

void put_frag(char* chunk, int np, int frag, int lane, int j, float v) {
    _Float16 hi, lo;
    r2l_split_f16(v, &hi, np == 2 ? &lo : nullptr);
    reinterpret_cast<_Float16*>(chunk + (size_t)(frag * np + 0) * R2L_FRAG_BYTES + lane * 16)[j] = hi;
    if (np == 2) reinterpret_cast<_Float16*>(chunk + (size_t)(frag * np + 1) * R2L_FRAG_BYTES + lane * 16)[j] = lo;
}

// One layer of the fragment stream: RT row tiles (16 outputs) x KS k-steps (32 inputs).
// weight(row, col) returns W[row][col] (0 outside), col_of(s, q, j) the input column of element
// j of lane quarter q of k-step s (or -1), bias(row) the bias.
void pack_layer(std::vector<char>& img, int np, int F0, int KS, int RT, float Sa,
                const std::function<float(int, int)>& weight, const std::function<int(int, int, int)>& col_of,
                const std::function<float(int)>& bias, float Sw, float* inv_scale_out) {
    const int CH = r2l_chunk_bytes(np);
    const int AUX = R2L_FRAGS * np * R2L_FRAG_BYTES;
    const float S = Sa * Sw;
    for (int u = 0; u < RT; ++u) {
        for (int ks = 0; ks < KS; ++ks) {
            const int fq = F0 + u * KS + ks;
            char* chunk = img.data() + (size_t)(fq / R2L_FRAGS) * CH;
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int col = col_of(ks, lane >> 4, j);
                    const float v = col < 0 ? 0.f : weight(16 * u + (lane & 15), col) * Sw;
                    put_frag(chunk, np, fq % R2L_FRAGS, lane, j, v);
                }
        }
        const int q0 = F0 + u * KS;
        float* aux = reinterpret_cast<float*>(img.data() + (size_t)(q0 / R2L_FRAGS) * CH + AUX);
        for (int i = 0; i < 16; ++i) aux[16 * nerf_aux_slot(q0) + i] = bias(16 * u + i) * S;
    }
    *inv_scale_out = 1.0f / S;
}

float max_scale(std::initializer_list<std::pair<const float*, size_t>> ts) {
    float m = 0.f;
    for (auto& t : ts)
        for (size_t i = 0; i < t.second; ++i) {
            float a = fabsf(t.first[i]);
            if (a > m && isfinite(a)) m = a;
        }
    return r2l_pow2_scale(&m, 1);
}


// FP16_FP8: the stream of the hand-scheduled layer chain (gen/nerf_gen.py; restated there as pack_teacher and compared
// byte for byte by tests/test_nerf_gen_cpu.py).  Eleven layers in execution order; a layer = RT row tiles of 16 outputs,
// cut into chunks of RPC row tiles; a chunk = 1 KiB pieces (64 lanes x 16 B), PW = ceil(pieces / 4) KiB per wave:
//   [k*KS + s]            fp16 hi fragment of main k-step s of the chunk's k-th row tile (UNSCALED weights)
//   [RPC*KS + k*NJ + j]   first 16 B/lane of bf6 operand j (NJ = KS/2: (term, t) = (0,0) (1,0) [(0,1) (1,1)])
//   then ceil(RPC*NJ/2) pieces holding the last 8 B/lane of operand k*NJ + j at byte (k*NJ + j) * 512
//   then per row tile and embedding k-step: the fp16 hi fragment, the fp16 lo fragment
// bf6 operands as in the R2L body (r2l_capi.hip): term 0 = (w - hi(w)) / 2^(e-16), term 1 = w / 2^(e-4), e = exponent
// of the layer's max|w|, element i of a lane = input feature r2l_mix_feat(t, lane>>4, i).  Behind the stream: the
// resident table, per layer NERF_CHAIN_AUX_LAYER bytes = bias x act_scale | the two E8M0 scale bytes per lane quarter.
struct ChainLayer { int ks, nx, rt, rpc, fan_out; };
const ChainLayer kChain[11] = {{0, 2, 16, 8, 256}, {8, 0, 16, 2, 256}, {8, 0, 16, 2, 256}, {8, 0, 16, 2, 256},
                               {8, 0, 16, 2, 256}, {8, 2, 16, 1, 256}, {8, 0, 16, 2, 256}, {8, 0, 16, 2, 256},
                               {8, 0, 17, 2, 257}, {8, 1, 8, 2, 128},  {4, 0, 1, 1, 3}};
// FP16X1 (no bf6 operands): twice the row tiles per chunk where a 32 KiB ring slot allows -- 44 chunks per tile instead of 80
const ChainLayer kChainX[11] = {{0, 2, 16, 8, 256}, {8, 0, 16, 4, 256}, {8, 0, 16, 4, 256}, {8, 0, 16, 4, 256},
                                {8, 0, 16, 4, 256}, {8, 2, 16, 2, 256}, {8, 0, 16, 4, 256}, {8, 0, 16, 4, 256},
                                {8, 0, 17, 4, 257}, {8, 1, 8, 2, 128},  {4, 0, 1, 1, 3}};
// FP16X3_ASM (gen/nerf_gen.py NERF_GEN_FMT=f16p3): per main k-step a hi AND a lo fragment of W x 2^k -- 84 chunks of <= 32 KiB
const ChainLayer kChainP3[11] = {{0, 2, 16, 8, 256}, {8, 0, 16, 2, 256}, {8, 0, 16, 2, 256}, {8, 0, 16, 2, 256},
                                 {8, 0, 16, 2, 256}, {8, 2, 16, 1, 256}, {8, 0, 16, 2, 256}, {8, 0, 16, 2, 256},
                                 {8, 0, 17, 2, 257}, {8, 1, 8, 1, 128},  {4, 0, 1, 1, 3}};

// FP16X3_ASM without the view branch (NERF_GEN_FMT=f16p3a): the trunk of kChainP3, then the alpha row alone (row tile 0; row tile 1 is
// padding: 68 chunks = 0 mod 4) -- for the coarse pass of renders whose caller does not take rgb0 (nerf_set_skip_rgb0)
const ChainLayer kChainP3A[9] = {{0, 2, 16, 8, 256}, {8, 0, 16, 2, 256}, {8, 0, 16, 2, 256}, {8, 0, 16, 2, 256},
                                 {8, 0, 16, 2, 256}, {8, 2, 16, 1, 256}, {8, 0, 16, 2, 256}, {8, 0, 16, 2, 256}, {8, 0, 2, 1, 1}};

// the chains with a second exit behind the density (NERF_GEN_FMT=f16p3s / mixs): layer 8 = the alpha row (one row tile, one chunk), layer 9 =
// the 256 feature rows, then views and rgb: 12 layers
const ChainLayer kChainP3S[12] = {{0, 2, 16, 8, 256}, {8, 0, 16, 2, 256}, {8, 0, 16, 2, 256}, {8, 0, 16, 2, 256}, {8, 0, 16, 2, 256}, {8, 2, 16, 1, 256},
                                  {8, 0, 16, 2, 256}, {8, 0, 16, 2, 256}, {8, 0, 1, 1, 1},   {8, 0, 16, 2, 256}, {8, 1, 8, 1, 128},  {4, 0, 1, 1, 3}};
const ChainLayer kChainS[12] = {{0, 2, 16, 8, 256}, {8, 0, 16, 2, 256}, {8, 0, 16, 2, 256}, {8, 0, 16, 2, 256}, {8, 0, 16, 2, 256}, {8, 2, 16, 1, 256},
                                {8, 0, 16, 2, 256}, {8, 0, 16, 2, 256}, {8, 0, 1, 1, 1},   {8, 0, 16, 2, 256}, {8, 1, 8, 2, 128},  {4, 0, 1, 1, 3}};

struct ChainSrc {   // one layer's parameters: main(row, k), emb(row, embedding column), bias(row); rows < fan_out
    std::function<float(int, int)> main, emb;
    std::function<float(int)> bias;
    bool is_pts;    // embedding k-steps: pts (nerf_pts_col) or view (nerf_view_col)
};

// fmt 1 (x1): the stream of R2L_PREC_FP16X1 -- the fp16 hi fragments and the embedding fragments only (NJ = 0), no scale bytes.
// fmt 2 (p3): the stream of R2L_PREC_FP16X3_ASM -- per row tile its KS hi fragments, then its KS lo fragments, of W x 2^k with
// max|w| 2^k in [2^12, 2^13) over the layer's main and embedding columns (lo = the fp16 rounding residual: a normal number thanks
// to the factor); the bias x act_scale x 2^k; at the scale bytes' place 2^-k as a float for the epilogue
// fmt 5 (p3s) / 6 (mixs): fmt 2 / 3 with the feature | alpha layer split (kChainP3S / kChainS: the alpha row first); the two halves' bf6 terms
// keep the unsplit layer's weight exponent, so that the split chain computes bit for bit what the unsplit one does
// fmt 4 (p3a): fmt 2 without the view branch (kChainP3A): nine layers, the last one the alpha row with its own 2^k
// fmt 3 (mix): the stream of R2L_PREC_FP16_MIX -- fmt 0 with layers L1 .. L<NERF_MIX_K> packed as in fmt 2 (hi | lo fragments of
// W x 2^k, 2^-k at the scale bytes' place) and 1.0f there for L0, whose epilogue hands L1 hi + lo sets with the three-pass form
int pack_chain(const std::vector<std::vector<float>>& w, float Sa, std::vector<char>& img, int fmt = 0) {
    const bool split = fmt == 5 || fmt == 6;
    const bool mix = fmt == 3 || fmt == 6, alpha_only = fmt == 4;
    const bool x1 = fmt == 1, all_p3 = fmt == 2 || alpha_only || fmt == 5;
    const size_t stream_bytes = fmt == 5 ? NERF_CHAINP3S_STREAM_BYTES : fmt == 6 ? NERF_CHAINMS_STREAM_BYTES : alpha_only ? NERF_CHAINP3A_STREAM_BYTES
                                : mix ? NERF_CHAINM_STREAM_BYTES : all_p3 ? NERF_CHAINP3_STREAM_BYTES
                                : (x1 ? NERF_CHAINX_STREAM_BYTES : NERF_CHAIN_STREAM_BYTES);
    const int n_layers = alpha_only ? 9 : split ? 12 : 11;
    img.assign(stream_bytes + NERF_CHAIN_AUX_BYTES, 0);
    auto mat = [&](int ti, int ncol, int col0) {
        const float* p = w[ti].data();
        return [p, ncol, col0](int r, int col) -> float { return p[(size_t)r * ncol + col0 + col]; };
    };
    auto vec = [&](int ti) {
        const float* p = w[ti].data();
        return [p](int r) -> float { return p[r]; };
    };
    ChainSrc src[12];
    src[0] = {nullptr, mat(0, 63, 0), vec(1), true};
    const int plain[6] = {1, 2, 3, 4, 6, 7};
    for (int li : plain) src[li] = {mat(2 * li, 256, 0), nullptr, vec(2 * li + 1), true};
    src[5] = {mat(10, 319, 63), mat(10, 319, 0), vec(11), true};     // cat([input_pts, h])  (model/nerf_raybased.py:385)
    {
        const float *fw = w[T_FEAT_W].data(), *aw = w[T_ALPHA_W].data(), *fb = w[T_FEAT_B].data(), *ab = w[T_ALPHA_B].data();
        src[8] = {[=](int r, int k) -> float { return r < 256 ? fw[(size_t)r * 256 + k] : aw[k]; }, nullptr,
                  [=](int r) -> float { return r < 256 ? fb[r] : ab[0]; }, true};
    }
    if (alpha_only) {
        const float *aw = w[T_ALPHA_W].data(), *ab = w[T_ALPHA_B].data();
        src[8] = {[=](int, int k) -> float { return aw[k]; }, nullptr, [=](int) -> float { return ab[0]; }, true};
    }
    src[9] = {mat(T_VIEWS_W, 283, 0), mat(T_VIEWS_W, 283, 256), vec(T_VIEWS_B), false};  // cat([feature, views]) (:390)
    src[10] = {mat(T_RGB_W, 128, 0), nullptr, vec(T_RGB_B), true};
    int fa_exp = 0;          // split chains: the weight exponent of the unsplit feature | alpha layer, shared by its two halves' bf6 terms,
    float fa_max = 0.f;      // and its max|w| (the three-pass halves' common 2^k): the split chain computes bit for bit what the unsplit one does
    if (split) {
        std::vector<float> all((size_t)257 * 256);
        for (int r = 0; r < 257; ++r)
            for (int k = 0; k < 256; ++k) all[(size_t)r * 256 + k] = src[8].main(r, k);
        fa_exp = r2l_layer_exponent(all.data(), all.size());
        for (float v : all) fa_max = fmaxf(fa_max, fabsf(v));
        const float *aw = w[T_ALPHA_W].data(), *ab = w[T_ALPHA_B].data();
        src[11] = src[10];
        src[10] = src[9];
        src[9] = {mat(T_FEAT_W, 256, 0), nullptr, vec(T_FEAT_B), true};
        src[8] = {[=](int, int k) -> float { return aw[k]; }, nullptr, [=](int) -> float { return ab[0]; }, true};
    }
    size_t chunk_off = 0;
    uint32_t* aux = reinterpret_cast<uint32_t*>(img.data() + stream_bytes);
    for (int li = 0; li < n_layers; ++li) {
        const ChainLayer& L = (fmt == 5 ? kChainP3S : fmt == 6 ? kChainS : alpha_only ? kChainP3A : all_p3 ? kChainP3 : (x1 ? kChainX : kChain))[li];
        const ChainSrc& S = src[li];
        const bool p3 = all_p3 || (mix && li >= 1 && li <= NERF_MIX_K);      // this layer's main k-steps run three fp16 passes
        const int nj = (x1 || p3) ? 0 : L.ks / 2, K = L.ks * 32;
        const int pieces = p3 ? L.rpc * L.ks * 2 + L.rpc * L.nx * 2 : L.rpc * L.ks + L.rpc * nj + (L.rpc * nj + 1) / 2 + L.rpc * L.nx * 2;
        float sw = 1.0f;
        if (p3) {
            float mx = 0.f;
            const int n_emb = S.emb ? (S.is_pts ? 63 : 27) : 0;
            for (int r = 0; r < L.fan_out; ++r) {
                for (int k = 0; k < K; ++k) mx = fmaxf(mx, fabsf(S.main(r, k)));
                for (int k = 0; k < n_emb; ++k) mx = fmaxf(mx, fabsf(S.emb(r, k)));
            }
            if (split && (li == 8 || li == 9)) mx = fa_max;        // the two halves of the feature | alpha layer keep the unsplit layer's 2^k
            sw = (mx > 0.f && isfinite(mx)) ? r2l_pow2_scale(&mx, 1) : 1.0f;
        }
        const size_t chunk_bytes = (size_t)((pieces + 3) / 4) * 4096;
        uint32_t* al = aux + (size_t)li * NERF_CHAIN_AUX_LAYER / 4;
        for (int r = 0; r < L.fan_out; ++r) {
            const float v = (float)((double)S.bias(r) * Sa * (double)sw);
            memcpy(&al[r], &v, 4);
        }
        if (p3 || (mix && li == 0)) {
            const float inv = 1.0f / sw;
            for (int q = 0; q < 4; ++q) memcpy(&al[NERF_CHAIN_AUX_SCALES / 4 + 4 * q], &inv, 4);
        }
        int el = 0, ew = 0;
        if (L.ks && !x1 && !p3) {
            std::vector<float> all((size_t)L.fan_out * K);
            for (int r = 0; r < L.fan_out; ++r)
                for (int k = 0; k < K; ++k) all[(size_t)r * K + k] = S.main(r, k);
            const int e = (split && (li == 8 || li == 9)) ? fa_exp : r2l_layer_exponent(all.data(), all.size());
            if (e < -12 || e > 6)
                return r2l_set_error(R2L_EINVAL, "teacher layer %d: max|w| = 2^%d is outside the range the fp16 + bf6 weight "
                                     "split covers (2^-12 .. 2^6); use R2L_PREC_FP16X3", li, e);
            el = e - 16;
            ew = e - 4;
            for (int q = 0; q < 4; ++q) {
                al[NERF_CHAIN_AUX_SCALES / 4 + 4 * q] = 0x01010101u * (uint32_t)(127 + el);
                al[NERF_CHAIN_AUX_SCALES / 4 + 4 * q + 1] = 0x01010101u * (uint32_t)(127 + ew);
            }
        }
        for (int u = 0; u < L.rt; ++u) {
            char* chunk = img.data() + chunk_off + (size_t)(u / L.rpc) * chunk_bytes;
            const int k = u % L.rpc;
            const int p_b6 = L.rpc * L.ks, p_b6b = p_b6 + L.rpc * nj, p_x = p3 ? L.rpc * L.ks * 2 : p_b6b + (L.rpc * nj + 1) / 2;
            for (int lane = 0; lane < 64; ++lane) {
                const int q = lane >> 4, row = 16 * u + (lane & 15);
                if (row >= L.fan_out) continue;
                for (int s = 0; s < L.ks && !p3; ++s) {
                    _Float16* ph = reinterpret_cast<_Float16*>(chunk + (size_t)(k * L.ks + s) * 1024 + lane * 16);
                    for (int j = 0; j < 8; ++j) ph[j] = (_Float16)S.main(row, r2l_kappa(s, q, j));
                }
                for (int s = 0; s < L.ks && p3; ++s) {
                    _Float16* ph = reinterpret_cast<_Float16*>(chunk + (size_t)(k * L.ks * 2 + s) * 1024 + lane * 16);
                    _Float16* pl = reinterpret_cast<_Float16*>(chunk + (size_t)(k * L.ks * 2 + L.ks + s) * 1024 + lane * 16);
                    for (int j = 0; j < 8; ++j) r2l_split_f16(S.main(row, r2l_kappa(s, q, j)) * sw, &ph[j], &pl[j]);
                }
                for (int j = 0; j < nj; ++j) {
                    const int term = j & 1, t = j >> 1;
                    uint64_t bits[3] = {0, 0, 0};
                    for (int i = 0; i < 32; ++i) {
                        const float wv = S.main(row, r2l_mix_feat(t, q, i));
                        const float hi = (float)(_Float16)wv;
                        const double v = term == 0 ? ldexp((double)wv - (double)hi, -el) : ldexp((double)wv, -ew);
                        const uint64_t code = r2l_f_to_bf6(v);
                        const int bit = 6 * i, wd = bit >> 6, sh = bit & 63;
                        bits[wd] |= code << sh;
                        if (sh > 58) bits[wd + 1] |= code >> (64 - sh);
                    }
                    memcpy(chunk + (size_t)(p_b6 + k * nj + j) * 1024 + lane * 16, bits, 16);
                    memcpy(chunk + (size_t)p_b6b * 1024 + (size_t)(k * nj + j) * 512 + lane * 8, &bits[2], 8);
                }
                for (int x = 0; x < L.nx; ++x) {
                    _Float16* ph = reinterpret_cast<_Float16*>(chunk + (size_t)(p_x + (k * L.nx + x) * 2) * 1024 + lane * 16);
                    _Float16* pl = ph + 512;
                    for (int j = 0; j < 8; ++j) {
                        const int col = S.is_pts ? nerf_pts_col(x, q, j) : nerf_view_col(q, j);
                        if (col < 0) continue;
                        r2l_split_f16(S.emb(row, col) * sw, &ph[j], &pl[j]);
                    }
                }
            }
        }
        chunk_off += (size_t)((L.rt + L.rpc - 1) / L.rpc) * chunk_bytes;
    }
    if (chunk_off != stream_bytes) return r2l_set_error(R2L_EINVAL, "internal: chain stream is %zu bytes", chunk_off);
    return R2L_OK;
}

}  // namespace

struct nerf_ctx {
    int H, W, N_samples, N_importance, white_bkgd, mode, n_cu;
    int mode_net[2] = {0, 0};   // precision of the coarse / the fine network's MLP launches (nerf_set_precision: both; nerf_set_precision_pair)
    int ndc = 0;            // render() projects the rays to NDC first (main.py:160-162)
    bool skip_rgb0 = false;    // nerf_set_skip_rgb0: the coarse pass of the render pipeline without its view branch when it runs fp16x3_asm
    bool split_scans = false;  // nerf_debug_set_split_scans: raw2outputs / sample_pdf / merge as three launches (A/B, parity tests)
    int x1_col_tiles = 4;      // nerf_debug_set_x1_col_tiles: 16-point column tiles per wave of the fp16-only chain (3 or 2 for the A/B)
    bool x1_stream_embed = true;   // nerf_debug_set_x1_stream_embed: the four-tile chain with its embedding in the stream (nerf_chain_emb_kernel)
    float ndc_near = 1.0f;
    double focal;
    float near_, far_, act_scale;
    std::vector<float> z_coarse, u;  // host copies
    float *d_zc = nullptr, *d_zmid = nullptr, *d_u = nullptr;
    float* d_zsort = nullptr;  // z_samples sorted (random uniforms only)
    PackedNet net[2];
    // per-call temporaries, grown on demand (sized for `cap` rays)
    int cap = 0;
    float *d_rays_o = nullptr, *d_rays_d = nullptr, *d_raw0 = nullptr, *d_w0 = nullptr, *d_zs = nullptr,
          *d_zall = nullptr, *d_raw = nullptr, *d_rgb0 = nullptr, *d_disp0 = nullptr, *d_acc0 = nullptr,
          *d_ndc_o = nullptr, *d_ndc_d = nullptr, *d_vdir = nullptr;
    // HIP-event timing of the MLP launches (nerf_timing_enable / nerf_kernel_time_ms), on the stream they are launched on
    bool timing = false;
    std::vector<hipEvent_t> ev;   // pairs
    int ev_used = 0;
};

static int ensure_alpha_img(nerf_ctx* c);

static void free_tmp(nerf_ctx* c) {
    float** ps[] = {&c->d_rays_o, &c->d_rays_d, &c->d_raw0, &c->d_w0, &c->d_zs, &c->d_zall, &c->d_raw,
                    &c->d_rgb0, &c->d_disp0, &c->d_acc0, &c->d_ndc_o, &c->d_ndc_d, &c->d_vdir, &c->d_zsort};
    for (auto p : ps)
        if (*p) {
            (void)hipFree(*p);
            *p = nullptr;
        }
    c->cap = 0;
}

static int ensure_tmp(nerf_ctx* c, int n) {
    if (n <= c->cap) return R2L_OK;
    free_tmp(c);
    const int S0 = c->N_samples, S1 = c->N_samples + c->N_importance;
    struct { float** p; size_t numel; } req[] = {
        {&c->d_rays_o, (size_t)n * 3}, {&c->d_rays_d, (size_t)n * 3}, {&c->d_raw0, (size_t)n * S0 * 4},
        {&c->d_w0, (size_t)n * S0},    {&c->d_zs, (size_t)n * c->N_importance}, {&c->d_zall, (size_t)n * S1},
        {&c->d_raw, (size_t)n * S1 * 4}, {&c->d_rgb0, (size_t)n * 3}, {&c->d_disp0, (size_t)n}, {&c->d_acc0, (size_t)n},
        {&c->d_ndc_o, (size_t)n * 3}, {&c->d_ndc_d, (size_t)n * 3}, {&c->d_vdir, (size_t)n * 3},
        {&c->d_zsort, (size_t)n * c->N_importance}};
    for (auto& r : req) {
        hipError_t e = hipMalloc((void**)r.p, r.numel * sizeof(float));
        if (e != hipSuccess) {
            free_tmp(c);
            return r2l_set_error(R2L_EHIP, "hipMalloc temporaries for %d rays: %s", n, hipGetErrorString(e));
        }
    }
    c->cap = n;
    return R2L_OK;
}

static int upload_sampling(nerf_ctx* c) {
    const int S = c->N_samples;
    std::vector<float> zmid(S - 1);
    for (int i = 0; i + 1 < S; ++i) {
        volatile float s = c->z_coarse[i + 1] + c->z_coarse[i];  // .5 * (z[1:] + z[:-1])  (main.py:722)
        volatile float m = .5f * s;
        zmid[i] = m;
    }
    struct { float** d; const float* h; size_t n; } up[] = {
        {&c->d_zc, c->z_coarse.data(), (size_t)S}, {&c->d_zmid, zmid.data(), (size_t)S - 1},
        {&c->d_u, c->u.data(), (size_t)c->N_importance}};
    for (auto& x : up) {
        if (!*x.d) {
            hipError_t e = hipMalloc((void**)x.d, x.n * sizeof(float));
            if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "hipMalloc: %s", hipGetErrorString(e));
        }
        hipError_t e = hipMemcpy(*x.d, x.h, x.n * sizeof(float), hipMemcpyHostToDevice);
        if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "hipMemcpy: %s", hipGetErrorString(e));
    }
    return R2L_OK;
}

int nerf_create(nerf_ctx** out, int H, int W, double focal, float near_, float far_, int N_samples, int N_importance,
                int multires, int multires_views, int white_bkgd, int precision_mode) {
    if (!out) return r2l_set_error(R2L_EINVAL, "out is NULL");
    *out = nullptr;
    if (multires != 10 || multires_views != 4)
        return r2l_set_error(R2L_EINVAL, "unsupported embedder multires=%d multires_views=%d (built for 10 / 4)",
                             multires, multires_views);
    // N_samples = 2 leaves sample_pdf a single bin and an empty cdf: the reference raises there too
    if (N_samples < 3 || N_samples > 64 || N_importance < 1 || N_samples + N_importance > 256)
        return r2l_set_error(R2L_EINVAL, "unsupported sampling N_samples=%d N_importance=%d (need 3..64 coarse, total <= 256)",
                             N_samples, N_importance);
    if (H <= 0 || W <= 0 || !(focal > 0)) return r2l_set_error(R2L_EINVAL, "bad geometry");
    if (!mode_ok(precision_mode)) return r2l_set_error(R2L_EINVAL, "bad precision_mode %d", precision_mode);
    int n_cu = 0;
    int rc = r2l_require_gfx950(&n_cu);
    if (rc) return rc;
    nerf_ctx* c = new nerf_ctx();
    c->H = H; c->W = W; c->focal = focal; c->near_ = near_; c->far_ = far_;
    c->N_samples = N_samples; c->N_importance = N_importance; c->white_bkgd = white_bkgd ? 1 : 0;
    c->mode = c->mode_net[0] = c->mode_net[1] = precision_mode; c->n_cu = n_cu; c->act_scale = 16.0f;
    c->z_coarse.resize(N_samples);
    r2l_z_vals(N_samples, near_, far_, c->z_coarse.data());  // main.py:676-678
    c->u.resize(N_importance);
    r2l_linspace01(N_importance, c->u.data());               // helpers:293
    rc = upload_sampling(c);
    if (rc) {
        nerf_destroy(c);
        return rc;
    }
    *out = c;
    return R2L_OK;
}

void nerf_destroy(nerf_ctx* c) {
    if (!c) return;
    free_tmp(c);
    for (auto& n : c->net)
        for (int m = 0; m < 8; ++m)
            if (n.d_img[m]) (void)hipFree(n.d_img[m]);
    for (auto& n : c->net) {
        if (n.d_img_alpha) (void)hipFree(n.d_img_alpha);
        if (n.d_img_exit) (void)hipFree(n.d_img_exit);
    }
    if (c->d_zc) (void)hipFree(c->d_zc);
    if (c->d_zmid) (void)hipFree(c->d_zmid);
    if (c->d_u) (void)hipFree(c->d_u);
    for (auto& e : c->ev) (void)hipEventDestroy(e);
    delete c;
}

int nerf_set_sampling(nerf_ctx* c, const float* z_coarse_host, int n_z, const float* u_host, int n_u) {
    if (!c) return r2l_set_error(R2L_EINVAL, "NULL ctx");
    if (z_coarse_host) {
        if (n_z != c->N_samples) return r2l_set_error(R2L_EINVAL, "expected %d coarse z values, got %d", c->N_samples, n_z);
        c->z_coarse.assign(z_coarse_host, z_coarse_host + n_z);
    }
    if (u_host) {
        if (n_u != c->N_importance) return r2l_set_error(R2L_EINVAL, "expected %d u values, got %d", c->N_importance, n_u);
        c->u.assign(u_host, u_host + n_u);
    }
    return upload_sampling(c);
}

static int upload_img(PackedNet& net, int mode, const std::vector<char>& img) {
    if (net.d_img[mode]) {
        (void)hipFree(net.d_img[mode]);
        net.d_img[mode] = nullptr;
    }
    hipError_t e = hipMalloc((void**)&net.d_img[mode], img.size());
    if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "hipMalloc image: %s", hipGetErrorString(e));
    e = hipMemcpy(net.d_img[mode], img.data(), img.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "hipMemcpy image: %s", hipGetErrorString(e));
    return R2L_OK;
}

static int build_net(nerf_ctx* c, PackedNet& net, int mode) {
    if (mode == R2L_PREC_FP16_FP8 || mode == R2L_PREC_FP16X1 || mode == R2L_PREC_FP16X3_ASM || mode == R2L_PREC_FP16_MIX) {  // the layer chain's own streams
        std::vector<char> img;
        int rc = pack_chain(net.host_w, c->act_scale, img, mode == R2L_PREC_FP16X1 ? 1 : (mode == R2L_PREC_FP16X3_ASM ? 2 : (mode == R2L_PREC_FP16_MIX ? 3 : 0)));
        return rc ? rc : upload_img(net, mode, img);
    }
    const int np = np_of(mode);
    const int CH = r2l_chunk_bytes(np);
    std::vector<char> img((size_t)NERF_CHUNKS * CH, 0);
    const float Sa = c->act_scale;
    auto& w = net.host_w;
    float* inv = net.inv_scale[mode];
    auto mat = [&](int ti, int ncol, int nrow) {
        const float* p = w[ti].data();
        return [p, ncol, nrow](int r, int col) -> float { return r < nrow ? p[(size_t)r * ncol + col] : 0.f; };
    };
    auto vec = [&](int ti, int nrow) {
        const float* p = w[ti].data();
        return [p, nrow](int r) -> float { return r < nrow ? p[r] : 0.f; };
    };
    auto kap = [](int ks, int h, int j) { return r2l_kappa(ks, h, j); };
    // L0
    pack_layer(img, np, NERF_F0_L0, 2, 16, Sa, mat(0, 63, 256), [](int ks, int q, int j) { return nerf_pts_col(ks, q, j); },
               vec(1, 256), r2l_pow2_scale(w[0].data(), w[0].size()), &inv[0]);
    // L1..L4, L6, L7
    const int plain[6] = {1, 2, 3, 4, 6, 7};
    for (int li : plain) {
        const int F0 = li <= 4 ? NERF_F0_L1 + 128 * (li - 1) : NERF_F0_L6 + 128 * (li - 6);
        pack_layer(img, np, F0, 8, 16, Sa, mat(2 * li, 256, 256), kap, vec(2 * li + 1, 256),
                   r2l_pow2_scale(w[2 * li].data(), w[2 * li].size()), &inv[li]);
    }
    // L5: reference input = cat[input_pts(63), h(256)]  (model/nerf_raybased.py:385)
    pack_layer(img, np, NERF_F0_L5, 10, 16, Sa, mat(10, 319, 256),
               [](int ks, int q, int j) {
                   if (ks < 8) return 63 + r2l_kappa(ks, q, j);
                   return nerf_pts_col(ks - 8, q, j);
               },
               vec(11, 256), r2l_pow2_scale(w[10].data(), w[10].size()), &inv[5]);
    // FA: rows 0..255 feature_linear, row 256 alpha_linear (row tile 16, row 0)
    {
        const float* fw = w[T_FEAT_W].data();
        const float* aw = w[T_ALPHA_W].data();
        const float* fb = w[T_FEAT_B].data();
        const float* ab = w[T_ALPHA_B].data();
        const float Sw = max_scale({{fw, w[T_FEAT_W].size()}, {aw, w[T_ALPHA_W].size()}});
        pack_layer(img, np, NERF_F0_FA, 8, 17, Sa,
                   [=](int r, int col) -> float { return r < 256 ? fw[(size_t)r * 256 + col] : (r == 256 ? aw[col] : 0.f); },
                   kap, [=](int r) -> float { return r < 256 ? fb[r] : (r == 256 ? ab[0] : 0.f); }, Sw, &inv[8]);
    }
    // V: reference input = cat[feature(256), input_views(27)]  (model/nerf_raybased.py:390)
    pack_layer(img, np, NERF_F0_V, 9, 8, Sa, mat(T_VIEWS_W, 283, 128),
               [](int ks, int q, int j) {
                   if (ks < 8) return r2l_kappa(ks, q, j);
                   const int cidx = nerf_view_col(q, j);
                   return cidx < 0 ? -1 : 256 + cidx;
               },
               vec(T_VIEWS_B, 128), r2l_pow2_scale(w[T_VIEWS_W].data(), w[T_VIEWS_W].size()), &inv[9]);
    // RGB
    pack_layer(img, np, NERF_F0_RGB, 4, 1, Sa, mat(T_RGB_W, 128, 3), kap, vec(T_RGB_B, 3),
               r2l_pow2_scale(w[T_RGB_W].data(), w[T_RGB_W].size()), &inv[10]);
    return upload_img(net, mode, img);
}

int nerf_load_weights(nerf_ctx* c, int which, const float* const* tensors, int n_tensors) {
    if (!c || !tensors) return r2l_set_error(R2L_EINVAL, "NULL argument");
    if (which != 0 && which != 1) return r2l_set_error(R2L_EINVAL, "which=%d (0 coarse, 1 fine)", which);
    if (n_tensors != 24) return r2l_set_error(R2L_EINVAL, "expected 24 tensors (NeRF D=8 W=256 use_viewdirs), got %d", n_tensors);
    PackedNet& net = c->net[which];
    net.host_w.clear();
    for (int i = 0; i < 24; ++i) {
        if (!tensors[i]) return r2l_set_error(R2L_EINVAL, "tensor %d is NULL", i);
        net.host_w.emplace_back(tensors[i], tensors[i] + kTensorNumel[i]);
    }
    for (int m = 0; m < 8; ++m)
        if (net.d_img[m]) {
            (void)hipFree(net.d_img[m]);
            net.d_img[m] = nullptr;
        }
    if (net.d_img_alpha) {
        (void)hipFree(net.d_img_alpha);
        net.d_img_alpha = nullptr;
    }
    if (net.d_img_exit) {
        (void)hipFree(net.d_img_exit);
        net.d_img_exit = nullptr;
        net.exit_mode = -1;
    }
    int rc = build_net(c, net, c->mode_net[which]);
    if (rc) return rc;
    net.loaded = true;
    return ensure_alpha_img(c);
}

int nerf_set_skip_rgb0(nerf_ctx* c, int on) {
    if (!c) return r2l_set_error(R2L_EINVAL, "NULL ctx");
    c->skip_rgb0 = on != 0;
    return ensure_alpha_img(c);
}

int nerf_set_precision(nerf_ctx* c, int mode) {
    if (!c) return r2l_set_error(R2L_EINVAL, "NULL ctx");
    if (!mode_ok(mode)) return r2l_set_error(R2L_EINVAL, "bad precision_mode %d", mode);
    return nerf_set_precision_pair(c, mode, mode);
}

int nerf_set_precision_pair(nerf_ctx* c, int coarse_mode, int fine_mode) {
    if (!c) return r2l_set_error(R2L_EINVAL, "NULL ctx");
    if (!mode_ok(coarse_mode) || !mode_ok(fine_mode)) return r2l_set_error(R2L_EINVAL, "bad precision_mode %d / %d", coarse_mode, fine_mode);
    const int modes[2] = {coarse_mode, fine_mode};
    for (int which = 0; which < 2; ++which) {
        PackedNet& n = c->net[which];
        if (n.loaded && !n.d_img[modes[which]]) {
            int rc = build_net(c, n, modes[which]);
            if (rc) return rc;
        }
        c->mode_net[which] = modes[which];
    }
    c->mode = fine_mode;
    return ensure_alpha_img(c);
}

// nerf_set_skip_rgb0: the coarse network's three-pass stream without its view branch (pack_chain fmt 4), packed outside the render path
static int ensure_alpha_img(nerf_ctx* c) {
    if (c->skip_rgb0) {      // ... and the fine network's stream with the second exit, for the mode it runs in
        PackedNet& f = c->net[1];
        const int fm = c->mode_net[1];
        if (f.loaded && (fm == R2L_PREC_FP16X3_ASM || fm == R2L_PREC_FP16_MIX) && f.exit_mode != fm) {
            std::vector<char> img;
            int rc = pack_chain(f.host_w, c->act_scale, img, fm == R2L_PREC_FP16_MIX ? 6 : 5);
            if (rc) return rc;
            if (f.d_img_exit) (void)hipFree(f.d_img_exit);
            f.d_img_exit = nullptr;
            f.exit_mode = -1;
            hipError_t e = hipMalloc((void**)&f.d_img_exit, img.size());
            if (e == hipSuccess) e = hipMemcpy(f.d_img_exit, img.data(), img.size(), hipMemcpyHostToDevice);
            if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "the fine network's stream with the second exit: %s", hipGetErrorString(e));
            f.exit_mode = fm;
        }
    }
    PackedNet& n = c->net[0];
    if (!c->skip_rgb0 || !n.loaded || n.d_img_alpha || c->mode_net[0] != R2L_PREC_FP16X3_ASM) return R2L_OK;
    std::vector<char> img;
    int rc = pack_chain(n.host_w, c->act_scale, img, 4);
    if (rc) return rc;
    hipError_t e = hipMalloc((void**)&n.d_img_alpha, img.size());
    if (e == hipSuccess) e = hipMemcpy(n.d_img_alpha, img.data(), img.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "the coarse network's stream without its view branch: %s", hipGetErrorString(e));
    return R2L_OK;
}

static int run_mlp(nerf_ctx* c, int which, const float* rays_o, const float* rays_d, const float* z, int z_stride,
                   int S, int n, float* raw, hipStream_t s, const float* viewdirs = nullptr, bool alpha_only = false, bool second_exit = false) {
    NerfMlpParams p;
    memset(&p, 0, sizeof p);
    p.viewdirs = viewdirs;
    const int mode = c->mode_net[which];
    // (the stream without the view branch is packed when the flag, the weights or the mode are set -- ensure_alpha_img --, never here:
    // a render allocates nothing and does not synchronise; without it the full kernel runs)
    alpha_only = alpha_only && mode == R2L_PREC_FP16X3_ASM && c->net[which].d_img_alpha != nullptr;
    // the second exit behind the density (fine network of a render whose caller drops the extras, no density noise): same rule
    second_exit = second_exit && !alpha_only && c->net[which].d_img_exit != nullptr && c->net[which].exit_mode == mode;
    p.wimg = alpha_only ? c->net[which].d_img_alpha : second_exit ? c->net[which].d_img_exit : c->net[which].d_img[mode];
    p.raw = raw;
    p.rays_o = rays_o;
    p.rays_d = rays_d;
    p.z = z;
    p.z_stride = z_stride;
    p.S = S;
    p.n_rays = n;
    p.n_pts = (long long)n * S;
    if (p.n_pts >= (1ll << 31))   // the kernels index points with 32 bits
        return r2l_set_error(R2L_EINVAL, "%d rays x %d samples: more than 2^31 points in one call; render fewer rows at a time", n, S);
    // the fp16-only chain: 256-point tiles (four column tiles per wave); with given view directions (NDC renders) 192: the four-tile
    // build has no register left to carry the next tile's directions across its asm block
    const int x1_tiles = c->x1_col_tiles;     // read once: tile size and kernel selection below must agree
    const int x1_nc = mode == R2L_PREC_FP16X1 ? ((viewdirs && x1_tiles == 4) ? 3 : x1_tiles) : 2;
    // the one-statement build addresses raw with 32-bit byte offsets (16 B per point) and z with 4 B per point
    const bool stream_embed = mode == R2L_PREC_FP16X1 && x1_nc == 4 && c->x1_stream_embed && !viewdirs && p.n_pts < (1ll << 28);
    {   // pt / S by multiplication (Granlund & Montgomery: exact for every 32-bit pt; gen/nerf_gen.py div_magic)
        const unsigned d = (unsigned)S;
        unsigned l = 0;
        while ((1ull << l) < d) ++l;
        p.div_magic = (unsigned)((((1ull << l) - d) << 32) / d + 1);
        p.div_sh1 = l < 1 ? l : 1;
        p.div_sh2 = l > 0 ? l - 1 : 0;
    }
    const int tile_pts = 64 * x1_nc;
    p.n_tiles = (int)((p.n_pts + tile_pts - 1) / tile_pts);
    p.act_scale = c->act_scale;
    p.neg1 = -1.0f;
    memcpy(p.inv_scale, c->net[which].inv_scale[mode], sizeof p.inv_scale);
    const int grid = balanced_grid(p.n_tiles, c->n_cu);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->timing) {
        if (c->ev_used + 2 > (int)c->ev.size())
            for (int i = 0; i < 2; ++i) {
                hipEvent_t ev;
                hipError_t er = hipEventCreate(&ev);
                if (er != hipSuccess) return r2l_set_error(R2L_EHIP, "hipEventCreate: %s", hipGetErrorString(er));
                c->ev.push_back(ev);
            }
        e0 = c->ev[c->ev_used];
        e1 = c->ev[c->ev_used + 1];
        c->ev_used += 2;
        (void)hipEventRecord(e0, s);
    }
    hipError_t e = nerf_launch_mlp(p, mode, grid, s, x1_nc, stream_embed, alpha_only, second_exit);
    if (c->timing) (void)hipEventRecord(e1, s);
    if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "nerf_mlp launch: %s", hipGetErrorString(e));
    return R2L_OK;
}

int nerf_timing_enable(nerf_ctx* c, int on) {
    if (!c) return r2l_set_error(R2L_EINVAL, "NULL ctx");
    c->timing = on != 0;
    return R2L_OK;
}

int nerf_kernel_time_ms(nerf_ctx* c, double* total_ms, int* n_launches, int reset) {
    if (!c || !total_ms || !n_launches) return r2l_set_error(R2L_EINVAL, "NULL argument");
    double tot = 0;
    for (int i = 0; i + 1 < c->ev_used; i += 2) {
        hipError_t e = hipEventSynchronize(c->ev[i + 1]);
        float ms = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]);
        if (e != hipSuccess) return r2l_set_error(R2L_EHIP, "event timing: %s", hipGetErrorString(e));
        tot += ms;
    }
    *total_ms = tot;
    *n_launches = c->ev_used / 2;
    if (reset) c->ev_used = 0;
    return R2L_OK;
}

#define HIPCHK(call, what)                                                                   \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) return r2l_set_error(R2L_EHIP, what ": %s", hipGetErrorString(e_)); \
    } while (0)

// Randomness of render_rays (training-time options of the reference; all null on the test path).  The caller draws
// the numbers exactly as the reference does (torch.rand / the pytest numpy streams) and hands them over:
//   z_coarse [n, N_samples]  jittered coarse depths, main.py:684-699 (perturb > 0)
//   u        [n, N_importance] sample_pdf's uniforms, helpers:298-307 (det = False)
//   noise0   [n, N_samples], noise1 [n, N_samples + N_importance]: randn * raw_noise_std, main.py:592-600
struct RenderOpts {
    const float* z_coarse = nullptr;
    const float* u = nullptr;
    const float* noise0 = nullptr;
    const float* noise1 = nullptr;
};

// render_rays (main.py:624-756) for n rays already in device memory
static int render_rays_dev(nerf_ctx* c, const float* rays_o, const float* rays_d, int n, float* rgb, float* disp,
                           float* acc, float* depth, hipStream_t s, const RenderOpts& o = RenderOpts()) {
    const int S0 = c->N_samples, NI = c->N_importance, S1 = S0 + NI;
    const float* vd = nullptr;
    if (c->ndc) {  // viewdirs from the world rays, then everything downstream sees the NDC rays (main.py:148-162)
        HIPCHK(nerf_launch_ndc_rays(rays_o, rays_d, n, c->H, c->W, c->focal, c->ndc_near, c->d_ndc_o, c->d_ndc_d,
                                    c->d_vdir, s), "ndc_rays");
        rays_o = c->d_ndc_o;
        rays_d = c->d_ndc_d;
        vd = c->d_vdir;
    }
    const float* zc = o.z_coarse ? o.z_coarse : c->d_zc;
    const int zc_stride = o.z_coarse ? S0 : 0;
    // coarse network_fn; with nerf_set_skip_rgb0 without its view branch: raw0 = (0, 0, 0, sigma), sigma bit for bit the full chain's
    int rc = run_mlp(c, 0, rays_o, rays_d, zc, zc_stride, S0, n, c->d_raw0, s, vd, c->skip_rgb0);
    if (rc) return rc;
    if (!o.u && !c->split_scans) {
        // deterministic test path: raw2outputs(coarse) + sample_pdf + merge as ONE launch (nerf_coarse_scan_kernel: the weights,
        // the cdf and the samples stay in LDS; bit-identical to the three launches below)
        HIPCHK(nerf_launch_coarse_scan(c->d_raw0, zc, zc_stride, rays_d, n, S0, c->white_bkgd, o.noise0, c->d_u, NI, c->d_rgb0,
                                       c->d_disp0, c->d_acc0, c->d_zs, c->d_zall, s), "coarse scan");
    } else {
        HIPCHK(nerf_launch_raw2outputs(c->d_raw0, zc, zc_stride, rays_d, n, S0, c->white_bkgd, c->d_rgb0, c->d_disp0,
                                       c->d_acc0, c->d_w0, nullptr, s, o.noise0), "raw2outputs(coarse)");
        // sample_pdf(z_vals_mid, weights[..., 1:-1], N_importance, det=(perturb == 0))   (main.py:722-728)
        if (o.z_coarse)
            HIPCHK(nerf_launch_sample_pdf(o.z_coarse, S0, 1, c->d_w0, S0, 1, n, S0 - 1, o.u ? o.u : c->d_u, o.u ? NI : 0, NI,
                                          c->d_zs, nullptr, nullptr, s), "sample_pdf");
        else
            HIPCHK(nerf_launch_sample_pdf(c->d_zmid, 0, 0, c->d_w0, S0, 1, n, S0 - 1, o.u ? o.u : c->d_u, o.u ? NI : 0, NI,
                                          c->d_zs, nullptr, nullptr, s), "sample_pdf");
        // z_vals = sort(cat(z_vals, z_samples))                                 (main.py:730-732): both rows ascending ->
        // rank merge; with random uniforms the samples are sorted first
        const float* zs_sorted = c->d_zs;
        if (o.u) {
            HIPCHK(nerf_launch_sort_rows(c->d_zs, n, NI, c->d_zsort, s), "sort z_samples");
            zs_sorted = c->d_zsort;
        }
        HIPCHK(nerf_launch_merge(zc, zc_stride, S0, zs_sorted, NI, n, c->d_zall, s), "merge");
    }
    // network_fine; with nerf_set_skip_rgb0 and no density noise on the chain with the second exit: the colours of workgroup tiles without a
    // positive density (weight 0 exactly) are not computed -- raw shows zeros there, every map is bit for bit the same
    rc = run_mlp(c, 1, rays_o, rays_d, c->d_zall, S1, S1, n, c->d_raw, s, vd, false, c->skip_rgb0 && !o.noise1);
    if (rc) return rc;
    HIPCHK(nerf_launch_raw2outputs(c->d_raw, c->d_zall, S1, rays_d, n, S1, c->white_bkgd, rgb, disp, acc, nullptr,
                                   depth, s, o.noise1), "raw2outputs(fine)");
    return R2L_OK;
}

int nerf_debug_set_x1_col_tiles(nerf_ctx* c, int n) {
    if (!c) return r2l_set_error(R2L_EINVAL, "NULL ctx");
    if (n < 2 || n > 4) return r2l_set_error(R2L_EINVAL, "column tiles per wave: 2, 3 or 4");
    c->x1_col_tiles = n;
    return R2L_OK;
}

int nerf_debug_set_x1_stream_embed(nerf_ctx* c, int on) {
    if (!c) return r2l_set_error(R2L_EINVAL, "NULL ctx");
    c->x1_stream_embed = on != 0;
    return R2L_OK;
}

int nerf_debug_set_split_scans(nerf_ctx* c, int on) {
    if (!c) return r2l_set_error(R2L_EINVAL, "NULL ctx");
    c->split_scans = on != 0;
    return R2L_OK;
}

int nerf_render_rays_ex(nerf_ctx* c, const float* rays_o_dev, const float* rays_d_dev, int n, const float* z_coarse_dev,
                        const float* u_dev, const float* noise0_dev, const float* noise1_dev, float* rgb_dev,
                        float* disp_dev, float* acc_dev, float* depth_dev, void* stream) {
    if (!c || !rays_o_dev || !rays_d_dev || !rgb_dev) return r2l_set_error(R2L_EINVAL, "NULL argument");
    if (!c->net[0].loaded || !c->net[1].loaded)
        return r2l_set_error(R2L_ESTATE, "nerf_render before nerf_load_weights of both networks");
    if (n < 0) return r2l_set_error(R2L_EINVAL, "n=%d", n);
    if (n == 0) return R2L_OK;
    int rc = ensure_tmp(c, n);
    if (rc) return rc;
    RenderOpts o;
    o.z_coarse = z_coarse_dev;
    o.u = u_dev;
    o.noise0 = noise0_dev;
    o.noise1 = noise1_dev;
    return render_rays_dev(c, rays_o_dev, rays_d_dev, n, rgb_dev, disp_dev, acc_dev, depth_dev, (hipStream_t)stream, o);
}

int nerf_render_rays(nerf_ctx* c, const float* rays_o_dev, const float* rays_d_dev, int n, float* rgb_dev,
                     float* disp_dev, float* acc_dev, float* depth_dev, void* stream) {
    if (!c || !rays_o_dev || !rays_d_dev || !rgb_dev) return r2l_set_error(R2L_EINVAL, "NULL argument");
    if (!c->net[0].loaded || !c->net[1].loaded)
        return r2l_set_error(R2L_ESTATE, "nerf_render before nerf_load_weights of both networks");
    if (n < 0) return r2l_set_error(R2L_EINVAL, "n=%d", n);
    if (n == 0) return R2L_OK;
    int rc = ensure_tmp(c, n);
    if (rc) return rc;
    return render_rays_dev(c, rays_o_dev, rays_d_dev, n, rgb_dev, disp_dev, acc_dev, depth_dev, (hipStream_t)stream);
}

int nerf_render(nerf_ctx* c, const float* c2w_host, int row_begin, int row_end, float* rgb_dev, float* disp_dev,
                float* acc_dev, float* depth_dev, void* stream) {
    if (!c || !c2w_host || !rgb_dev) return r2l_set_error(R2L_EINVAL, "NULL argument");
    if (!c->net[0].loaded || !c->net[1].loaded)
        return r2l_set_error(R2L_ESTATE, "nerf_render before nerf_load_weights of both networks");
    if (row_begin < 0 || row_end > c->H || row_begin >= row_end)
        return r2l_set_error(R2L_EINVAL, "bad row range [%d,%d) for H=%d", row_begin, row_end, c->H);
    const int n = (row_end - row_begin) * c->W;
    int rc = ensure_tmp(c, n);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(nerf_launch_get_rays(c2w_host, c->W, (float)(c->W * .5), (float)(c->H * .5), (float)c->focal,
                                row_begin * c->W, n, c->d_rays_o, c->d_rays_d, s), "get_rays");
    return render_rays_dev(c, c->d_rays_o, c->d_rays_d, n, rgb_dev, disp_dev, acc_dev, depth_dev, s);
}

int nerf_set_ndc(nerf_ctx* c, int on, float ndc_near) {
    if (!c) return r2l_set_error(R2L_EINVAL, "NULL ctx");
    if (on && !(ndc_near > 0)) return r2l_set_error(R2L_EINVAL, "ndc_near=%g", ndc_near);
    c->ndc = on ? 1 : 0;
    c->ndc_near = ndc_near;
    return R2L_OK;
}

int nerf_ndc_rays(int H, int W, double focal, float near_, const float* rays_o_dev, const float* rays_d_dev, int n,
                  float* out_o_dev, float* out_d_dev, void* stream) {
    if (!rays_o_dev || !rays_d_dev || !out_o_dev || !out_d_dev || n < 0 || H <= 0 || W <= 0 || !(focal > 0))
        return r2l_set_error(R2L_EINVAL, "bad argument to nerf_ndc_rays");
    int rc = r2l_require_gfx950(nullptr);
    if (rc) return rc;
    if (n == 0) return R2L_OK;
    HIPCHK(nerf_launch_ndc_rays(rays_o_dev, rays_d_dev, n, H, W, focal, near_, out_o_dev, out_d_dev, nullptr,
                                (hipStream_t)stream), "ndc_rays");
    return R2L_OK;
}

int nerf_last_extras(nerf_ctx* c, const float** rgb0, const float** z_samples, const float** z_vals,
                     const float** raw_fine) {
    if (!c || c->cap == 0) return r2l_set_error(R2L_ESTATE, "no render has run on this context");
    if (rgb0) *rgb0 = c->d_rgb0;
    if (z_samples) *z_samples = c->d_zs;
    if (z_vals) *z_vals = c->d_zall;
    if (raw_fine) *raw_fine = c->d_raw;
    return R2L_OK;
}

int nerf_copy_extras(nerf_ctx* c, int n, float* rgb0_dev, float* z_samples_dev, float* z_vals_dev, float* raw_dev,
                      void* stream) {
    if (!c || c->cap == 0) return r2l_set_error(R2L_ESTATE, "no render has run on this context");
    if (n < 0 || n > c->cap) return r2l_set_error(R2L_EINVAL, "n=%d exceeds the last render (%d rays)", n, c->cap);
    const int S1 = c->N_samples + c->N_importance;
    hipStream_t s = (hipStream_t)stream;
    if (rgb0_dev && c->skip_rgb0 && c->mode_net[0] == R2L_PREC_FP16X3_ASM)
        return r2l_set_error(R2L_ESTATE, "rgb0 was not computed: nerf_set_skip_rgb0 is on (the coarse pass ran without its view branch)");
    struct { float* dst; const float* src; size_t numel; } cp[] = {
        {rgb0_dev, c->d_rgb0, (size_t)n * 3}, {z_samples_dev, c->d_zs, (size_t)n * c->N_importance},
        {z_vals_dev, c->d_zall, (size_t)n * S1}, {raw_dev, c->d_raw, (size_t)n * S1 * 4}};
    for (auto& x : cp)
        if (x.dst) HIPCHK(hipMemcpyAsync(x.dst, x.src, x.numel * sizeof(float), hipMemcpyDeviceToDevice, s), "copy extras");
    return R2L_OK;
}

// the rest of render_rays' return set (main.py:743-750): disp0, acc0 of the coarse pass and
// z_std = torch.std(z_samples, dim=-1, unbiased=False); any pointer may be null
int nerf_copy_extras0(nerf_ctx* c, int n, float* disp0_dev, float* acc0_dev, float* z_std_dev, void* stream) {
    if (!c || c->cap == 0) return r2l_set_error(R2L_ESTATE, "no render has run on this context");
    if (n < 0 || n > c->cap) return r2l_set_error(R2L_EINVAL, "n=%d exceeds the last render (%d rays)", n, c->cap);
    hipStream_t s = (hipStream_t)stream;
    if (disp0_dev) HIPCHK(hipMemcpyAsync(disp0_dev, c->d_disp0, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s), "copy disp0");
    if (acc0_dev) HIPCHK(hipMemcpyAsync(acc0_dev, c->d_acc0, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s), "copy acc0");
    if (z_std_dev && n > 0) HIPCHK(nerf_launch_row_std(c->d_zs, n, c->N_importance, z_std_dev, s), "z_std");
    return R2L_OK;
}

int nerf_get_rays(int H, int W, double focal, const float* c2w_host, int row_begin, int row_end, float* rays_o_dev,
                  float* rays_d_dev, void* stream) {
    if (!c2w_host || !rays_o_dev || !rays_d_dev || row_begin < 0 || row_end > H || row_begin >= row_end)
        return r2l_set_error(R2L_EINVAL, "bad argument to nerf_get_rays");
    int rc = r2l_require_gfx950(nullptr);
    if (rc) return rc;
    HIPCHK(nerf_launch_get_rays(c2w_host, W, (float)(W * .5), (float)(H * .5), (float)focal, row_begin * W,
                                (row_end - row_begin) * W, rays_o_dev, rays_d_dev, (hipStream_t)stream), "get_rays");
    return R2L_OK;
}

// run_network (main.py:65-87) on explicit z values: raw [n, S, 4]
long long nerf_debug_pack_chain_host(const float* const* tensors, int n_tensors, int fmt, char* out, long long cap, long long* offs) {
    if (fmt < 0 || fmt > 6)
        return r2l_set_error(R2L_EINVAL, "chain stream format %d (0 fp16 + bf6 terms, 1 fp16 only, 2 hi | lo, 3 mix, 4 hi | lo without the view branch, "
                             "5 / 6: 2 / 3 with the second exit behind the density)", fmt);
    if (!tensors || n_tensors != 24) return r2l_set_error(R2L_EINVAL, "expected 24 tensors");
    std::vector<std::vector<float>> w;
    for (int i = 0; i < 24; ++i) {
        if (!tensors[i]) return r2l_set_error(R2L_EINVAL, "tensor %d is NULL", i);
        w.emplace_back(tensors[i], tensors[i] + kTensorNumel[i]);
    }
    std::vector<char> img;
    int rc = pack_chain(w, 16.0f, img, fmt);
    if (rc) return rc;
    if (offs) offs[0] = fmt == 6 ? NERF_CHAINMS_STREAM_BYTES : fmt == 5 ? NERF_CHAINP3S_STREAM_BYTES : fmt == 4 ? NERF_CHAINP3A_STREAM_BYTES : fmt == 3 ? NERF_CHAINM_STREAM_BYTES : fmt == 2 ? NERF_CHAINP3_STREAM_BYTES : (fmt == 1 ? NERF_CHAINX_STREAM_BYTES : NERF_CHAIN_STREAM_BYTES);
    if (out && cap > 0) memcpy(out, img.data(), (size_t)(cap < (long long)img.size() ? cap : (long long)img.size()));
    return (long long)img.size();
}

int nerf_run_network(nerf_ctx* c, int which, const float* rays_o_dev, const float* rays_d_dev, const float* z_dev,
                     int z_stride, int S, int n, float* raw_dev, void* stream) {
    if (!c || !rays_o_dev || !rays_d_dev || !z_dev || !raw_dev) return r2l_set_error(R2L_EINVAL, "NULL argument");
    if (which != 0 && which != 1) return r2l_set_error(R2L_EINVAL, "which=%d", which);
    if (!c->net[which].loaded) return r2l_set_error(R2L_ESTATE, "network %d not loaded", which);
    if (S < 1 || n < 0 || (z_stride != 0 && z_stride < S)) return r2l_set_error(R2L_EINVAL, "bad S/n/z_stride");
    if (n == 0) return R2L_OK;
    return run_mlp(c, which, rays_o_dev, rays_d_dev, z_dev, z_stride, S, n, raw_dev, (hipStream_t)stream);
}

int nerf_raw2outputs_noise(const float* raw, const float* z, const float* rays_d, const float* noise, int n, int S,
                           int white_bkgd, float* rgb, float* disp, float* acc, float* weights, float* depth, void* stream) {
    if (!raw || !z || !rays_d || n < 0 || S < 1 || S > 256)
        return r2l_set_error(R2L_EINVAL, "bad argument to nerf_raw2outputs (S must be 1..256)");
    int rc = r2l_require_gfx950(nullptr);
    if (rc) return rc;
    if (n == 0) return R2L_OK;
    HIPCHK(nerf_launch_raw2outputs(raw, z, S, rays_d, n, S, white_bkgd, rgb, disp, acc, weights, depth,
                                   (hipStream_t)stream, noise), "raw2outputs");
    return R2L_OK;
}

int nerf_raw2outputs(const float* raw, const float* z, const float* rays_d, int n, int S, int white_bkgd, float* rgb,
                     float* disp, float* acc, float* weights, float* depth, void* stream) {
    return nerf_raw2outputs_noise(raw, z, rays_d, nullptr, n, S, white_bkgd, rgb, disp, acc, weights, depth, stream);
}

int nerf_sample_pdf_ex(const float* bins, const float* weights, int n, int n_bins, const float* u_dev, int u_per_ray, int N,
                       float* samples, float* cdf_out_dev, int* inds_out_dev, void* stream) {
    if (!bins || !weights || !samples || n < 0 || n_bins < 2 || n_bins > 64 || N < 1)
        return r2l_set_error(R2L_EINVAL, "bad argument to nerf_sample_pdf (n_bins must be 2..64)");
    if (u_per_ray && !u_dev) return r2l_set_error(R2L_EINVAL, "u_per_ray without u");
    int rc = r2l_require_gfx950(nullptr);
    if (rc) return rc;
    if (n == 0) return R2L_OK;
    HIPCHK(nerf_launch_sample_pdf(bins, n_bins, 0, weights, n_bins - 1, 0, n, n_bins, u_dev, u_per_ray ? N : 0, N, samples,
                                  cdf_out_dev, inds_out_dev, (hipStream_t)stream), "sample_pdf");
    return R2L_OK;
}

// u = torch.linspace(0, 1, N) by the scalar formula, evaluated in the kernel (stream-ordered, nothing allocated)
int nerf_sample_pdf(const float* bins, const float* weights, int n, int n_bins, int N, float* samples, void* stream) {
    return nerf_sample_pdf_ex(bins, weights, n, n_bins, nullptr, 0, N, samples, nullptr, nullptr, stream);
}

int nerf_sample_pdf_u(const float* bins, const float* weights, int n, int n_bins, const float* u_dev, int N,
                      float* samples, void* stream) {
    if (!u_dev) return r2l_set_error(R2L_EINVAL, "bad argument to nerf_sample_pdf_u (u is NULL)");
    return nerf_sample_pdf_ex(bins, weights, n, n_bins, u_dev, 0, N, samples, nullptr, nullptr, stream);
}

int nerf_merge_sorted(const float* a, int na, const float* b, int nb, int n, float* out, void* stream) {
    if (!a || !b || !out || n < 0 || na < 0 || nb < 0 || na > 256 || nb > 256)
        return r2l_set_error(R2L_EINVAL, "bad argument to nerf_merge_sorted (row lengths must be <= 256)");
    int rc = r2l_require_gfx950(nullptr);
    if (rc) return rc;
    if (n == 0) return R2L_OK;
    HIPCHK(nerf_launch_merge(a, na, na, b, nb, n, out, (hipStream_t)stream), "merge");
    return R2L_OK;
}
