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
    float c2w_host[12];
    const float* z;        // device [16]: PointSampler.z_vals (model/nerf_raybased.py:88-90); a pointer, not an
                           // array: dynamic indexing into a by-value kernarg array spills the struct to scratch
    float focal, half_w, half_h, act_scale;
    float neg1;  // -1.0f (kept opaque to the compiler: selects v_fma_mix for the hi/lo residuals)
    int W, pix_begin, rays_per_pose, n_rays, n_tiles, n_block, use_residual, chunks_per_tile;
};

hipError_t r2l_launch_resmlp(const R2LParams& p, int mode, int grid, hipStream_t stream);  // mode: R2L_PREC_*
hipError_t r2l_launch_sample_embed(const R2LParams& p, float* pts_out, float* emb_out,
                                   hipStream_t stream);
hipError_t r2l_launch_embed(const float* x, long long total, int L, float* emb_out,
                            hipStream_t stream);
