// Kernel parameter block + launcher prototypes shared by r2l_kernels.hip and r2l_capi.hip.
#pragma once
#include <hip/hip_runtime.h>

struct R2LParams {
    const char* wimg;      // packed chunk stream (device), mode-specific
    float* rgb;            // [n_rays, 3] out
    const float* c2w;      // device [n_pose, 3, 4] or nullptr (then c2w_host)
    const float* rays_o;   // device [n_rays, 3] or nullptr (then camera rays)
    const float* rays_d;
    float* scratch;        // [grid, 4 waves, 32, 64 lanes, 4] f32: head output kept for the global skip
    float* xbuf;           // head-only launch: [n_tiles, 4 waves, 32, 64 lanes, 4] f32 head output
    int tile_begin;        // first ray tile of this launch within the call (ray index = (tile_begin + tile) * 128 + ...)
    float c2w_host[12];
    const float* z;        // device [16]: PointSampler.z_vals (model/nerf_raybased.py:88-90); a pointer, not an
                           // array: dynamic indexing into a by-value kernarg array spills the struct to scratch
    float focal, half_w, half_h, act_scale;
    float neg1;  // -1.0f (kept opaque to the compiler: selects v_fma_mix for the hi/lo residuals)
    int W, pix_begin, rays_per_pose, n_rays, n_tiles, n_block, use_residual, chunks_per_tile;
    unsigned* range;       // head launch: [0] <- atomicMax of the f32 bits of max h0 (act_scale domain) over its rays; nullable
    // activations of the compiler-scheduled kernels as slopes s: act(v) = max(v, s v) -- 0 relu, 0.01 LeakyReLU, 1 none
    // (model/nerf_raybased.py:468-476 get_activation): head (args.act), inside a ResMLP block (trial.inact), behind it (trial.outact)
    float act_head, act_in, act_out;
    float block_resid;     // 1: x = x + W2 h + b2 (ResMLP, trial.body_arch = resmlp); 0: x = W2 h + b2 (pairs of plain layers, = mlp)
};

hipError_t r2l_launch_resmlp(const R2LParams& p, int mode, int grid, hipStream_t stream);  // mode: R2L_PREC_*
// head layer -> p.xbuf; x3: the three-fp16-pass build (R2L_PREC_FP16X3_ASM, stream of pack_head_v1(f16))
hipError_t r2l_launch_head(const R2LParams& p, int grid, hipStream_t stream, int x3 = 0);

// hand-scheduled body (r2l_body.hip): x <- ResMLP blocks(x) on the register image written by the head launch
struct R2LBodyParams {
    const char* wimg;   // body stream: n_block * 16 chunks of 28 KiB (r2l_capi.hip pack_body_v3)
    const char* aux;    // n_block aux blocks of 4 KiB (bias of layer 1 | E8M0 scales)
    const float* xin;   // [n_tiles, 4, 32, 64, 4] f32
    float* xout;        // x image out; unused when rgb != nullptr
    int n_tiles, n_block;
    // fused tail (networks with the global skip): rgb rows of the call, nullptr = write the x image instead
    float* rgb;
    const float* tail;  // 4 KiB: [3, 256] tail weight / act_scale | 2 x (3 folded biases, 0)
    int n_rays, tile_begin;
    // range guard: 2 n_block maxima (f32 bits, act_scale domain) of the operand sets IN_0, H_0, IN_1, ... over every ray of
    // the launch; nullptr = the plain kernel (r2l_body.hip: r2l_body_guard_kernel)
    unsigned* gstats;
    int e4m3;           // 0: bf6 correction terms (r2l_body_kernel, 28 KiB chunks); 1: e4m3 (r2l_body8_kernel, 32 KiB chunks);
                        // 2: three fp16 passes (r2l_bodyx_kernel, 32 KiB chunks: R2L_PREC_FP16X3_ASM; no guard build)
};
struct R2LTailParams {
    const float* xa;    // head output (global skip), may be nullptr
    const float* xb;    // body output (or head output when n_block == 0)
    const float* wt;    // [3, 256] tail weight / act_scale, then 3 folded biases
    float* rgb;         // [n_rays, 3]
    int n_tiles, tile_begin, n_rays;
};
hipError_t r2l_launch_body(const R2LBodyParams& p, int grid, hipStream_t stream);
hipError_t r2l_launch_tail(const R2LTailParams& p, hipStream_t stream);
// activation exponents of the body from the head output of n_tiles ray tiles at xa (register image): stream-ordered,
// results land in the aux blocks at `aux` (n_block x R2L_BODY_AUX_BYTES); wcal: per block W1^T | b1' | W2^T in fp32
#define R2L_CALIB_TILES 8   // ray tiles (1,024 rays) the activation-range measurement samples from one call
// accumulate != 0: keep the maxima already in stats (earlier calls with fewer tiles than the sample)
// exps: the same 2 n_block + 1 exponents as one contiguous int array (what the host reads back, one copy)
hipError_t r2l_launch_calib(const float* xa, const float* wcal, int n_block, int n_tiles, float act_scale, unsigned* stats,
                            char* aux, int* exps, int accumulate, hipStream_t stream);
// exponents from the range guard's maxima (gstats: 2 n_block sets, act_scale domain) and the head's h0 maximum (range[0])
hipError_t r2l_launch_recalibrate(const unsigned* gstats, const unsigned* range, int n_block, char* aux, int* exps, hipStream_t stream);
// r2l_set_act_exponents: exps (device, contiguous) -> the aux blocks
hipError_t r2l_launch_spread_exponents(const int* exps, int n_block, char* aux, hipStream_t stream);
// aux blocks [first, first + count) of the bf6 stream for a launch over just those blocks (R2L_PREC_FP16_SPLIT), made on the device
hipError_t r2l_launch_split_aux(const char* aux, int first, int count, char* dst, hipStream_t stream);
int r2l_body_lds_bytes();
int r2l_body_guard_max_blocks(int e4m3);   // largest n_block whose 2 n_block maxima rows fit the LDS beside the ring
hipError_t r2l_launch_sample_embed(const R2LParams& p, float* pts_out, float* emb_out,
                                   hipStream_t stream);
hipError_t r2l_launch_embed(const float* x, long long total, int L, float* emb_out,
                            hipStream_t stream);
