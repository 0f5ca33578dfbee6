// NeRF-teacher kernels: parameter blocks + launcher prototypes (nerf_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "nerf_common.h"

struct NerfMlpParams {
    const char* wimg;       // packed fragment stream of one network (device), mode-specific
    float* raw;             // [n_pts, 4] out: rgb(3), sigma(1)   (NeRF.forward output order)
    const float* rays_o;    // [n_rays, 3]
    const float* rays_d;    // [n_rays, 3]
    const float* viewdirs;  // [n_rays, 3] unit view directions, or NULL: rays_d / ||rays_d||
    const float* z;         // [n_rays, S] or, when z_stride == 0, one shared row [S]
    int z_stride;
    int S;                  // samples per ray
    int n_rays;
    long long n_pts;        // n_rays * S
    int n_tiles;            // ceil(n_pts / 128)
    float act_scale;
    float neg1;             // -1.0f, opaque to the compiler (r2l_device.h pack_lo)
    float inv_scale[NERF_N_SCALES];  // per layer: 1 / (act_scale * weight_scale)
    // point -> (ray, sample) without a division in the kernel: pt / S = (t + ((pt - t) >> div_sh1)) >> div_sh2, t = mulhi(pt, div_magic)
    unsigned div_magic, div_sh1, div_sh2;
};

// mode: R2L_PREC_*; x1_col_tiles (FP16X1 only): 16-point column tiles per wave, 2 or 3 -- p.n_tiles must count tiles of 64 x that many points
// stream_embed (FP16X1, four column tiles, no given view directions): nerf_chain_emb_kernel, the chain as one statement that loads its
// rays, computes the next tile's embedding under the MFMAs and stores raw itself (needs p.div_* and n_pts < 2^28)
hipError_t nerf_launch_mlp(const NerfMlpParams& p, int mode, int grid, hipStream_t stream, int x1_col_tiles = 2, bool stream_embed = false,
                           bool alpha_only = false, bool second_exit = false);

// rays of rows [row_begin,row_end) of one frame (utils/run_nerf_raybased_helpers.py:231-257)
hipError_t nerf_launch_get_rays(const float* c2w12_host, int W, float half_w, float half_h, float focal,
                                int pix_begin, int n, float* rays_o, float* rays_d, hipStream_t stream);

// ndc_rays (utils/run_nerf_raybased_helpers.py:260-279) of n rays; viewdirs (optional) receives
// rays_d / ||rays_d|| of the input rays; out_o / out_d may be NULL when only viewdirs is wanted
hipError_t nerf_launch_ndc_rays(const float* rays_o, const float* rays_d, int n, int H, int W, double focal, float near_,
                                float* out_o, float* out_d, float* viewdirs, hipStream_t stream);

// raw [n,S,4], z [n,S] (z_stride 0 = shared row), rays_d [n,3]; any output may be null; noise [n,S] or null is
// added to the density before the relu (raw_noise_std > 0, main.py:592-600)
hipError_t nerf_launch_raw2outputs(const float* raw, const float* z, int z_stride, const float* rays_d, int n,
                                   int S, int white_bkgd, float* rgb, float* disp, float* acc, float* weights,
                                   float* depth, hipStream_t stream, const float* noise = nullptr);
// bins [n,n_bins] (stride 0 = shared; bins_are_z: rows of n_bins + 1 depths, bins = their midpoints), weights
// [n, w_stride] using columns [w_off, w_off + n_bins - 1); u null (= linspace(0,1,N)), [N] (u_stride 0) or [n,N];
// cdf_out [n,n_bins] / inds_out [n,N] optional taps
hipError_t nerf_launch_sample_pdf(const float* bins, int bins_stride, int bins_are_z, const float* weights, int w_stride,
                                  int w_off, int n, int n_bins, const float* u, int u_stride, int N, float* samples,
                                  float* cdf_out, int* inds_out, hipStream_t stream);
// raw2outputs(coarse) + sample_pdf(z_mid, weights[..., 1:-1], N, det) + merge in one launch (main.py:705-732 with perturb = 0):
// raw [n,S,4], z [n,S] (z_stride 0 = shared row), S <= 64; u null (= linspace) or one shared row [N]; writes the coarse maps
// (any may be null), samples [n,N] and z_all [n, S + N]; bit-identical to the three stand-alone launches
hipError_t nerf_launch_coarse_scan(const float* raw, const float* z, int z_stride, const float* rays_d, int n, int S, int white_bkgd,
                                   const float* noise, const float* u, int N, float* rgb, float* disp, float* acc, float* samples,
                                   float* z_all, hipStream_t stream);
// out[i, :] = sort(x[i, :N]) ascending, N <= 256
hipError_t nerf_launch_sort_rows(const float* x, int n, int N, float* out, hipStream_t stream);
// out[i] = std(x[i, :N], unbiased=False)
hipError_t nerf_launch_row_std(const float* x, int n, int N, float* out, hipStream_t stream);
// a [n,na] (stride 0 = shared) and b [n,nb] ascending -> out [n, na+nb] ascending
hipError_t nerf_launch_merge(const float* a, int a_stride, int na, const float* b, int nb, int n, float* out,
                             hipStream_t stream);
