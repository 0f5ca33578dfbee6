/* A torch-free client of the C-ABI (include/r2l_hip.h): load an R2L network from a flat
 * float32 file, render one frame with the fused HIP kernel, write the RGB floats.
 *
 *   weights file : the 4 + 4*n_block state_dict tensors back to back, [out,in] row-major f32, in
 *                  the order head.0.{weight,bias}, body.i.body.{0,2}.{weight,bias}, tail.0.{weight,bias}
 *   pose file    : 12 floats, c2w[:3,:4] row-major
 *   output file  : H*W*3 floats
 *
 * Build: `make -C efficient-nerf_amd/csrc example` (plain gcc against libr2l_hip.so + the HIP runtime)
 * Run:
 *   examples/r2l_render weights.bin pose.bin out.bin H W focal n_block [precision]
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>

#include "r2l_hip.h"

static float* read_floats(const char* path, size_t n) {
    FILE* f = fopen(path, "rb");
    if (!f) {
        fprintf(stderr, "cannot open %s\n", path);
        exit(2);
    }
    float* p = (float*)malloc(n * sizeof(float));
    if (fread(p, sizeof(float), n, f) != n) {
        fprintf(stderr, "%s: expected %zu floats\n", path, n);
        exit(2);
    }
    fclose(f);
    return p;
}

#define CHECK(call)                                                          \
    do {                                                                     \
        int rc_ = (call);                                                    \
        if (rc_ != R2L_OK) {                                                 \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, r2l_last_error()); \
            return 1;                                                        \
        }                                                                    \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 8) {
        fprintf(stderr, "usage: %s weights.bin pose.bin out.bin H W focal n_block [precision 0|1|2]\n", argv[0]);
        return 2;
    }
    const int H = atoi(argv[4]), W = atoi(argv[5]), n_block = atoi(argv[7]);
    const double focal = atof(argv[6]);
    const int prec = argc > 8 ? atoi(argv[8]) : R2L_PREC_FP16X3;
    if (r2l_device_count() < 1) {
        fprintf(stderr, "no gfx950 device\n");
        return 3;
    }
    const int nt = 4 + 4 * n_block;
    size_t total = (size_t)256 * 1008 + 256 + (size_t)n_block * 2 * (256 * 256 + 256) + 3 * 256 + 3;
    float* blob = read_floats(argv[1], total);
    const float** tensors = (const float**)malloc(nt * sizeof(float*));
    size_t off = 0;
    for (int i = 0; i < nt; ++i) {
        tensors[i] = blob + off;
        if (i == 0) off += (size_t)256 * 1008;
        else if (i == nt - 2) off += 3 * 256;
        else if (i == nt - 1) off += 3;
        else off += (i % 2 == 0) ? (size_t)256 * 256 : 256;  /* i even: weight (i = 0 handled), odd: bias */
    }
    float* pose = read_floats(argv[2], 12);

    r2l_ctx* ctx = NULL;
    CHECK(r2l_create(&ctx, H, W, focal, 2.0f, 6.0f, 16, 10, 256, n_block, 1, prec));
    CHECK(r2l_load_weights(ctx, tensors, nt));
    float* rgb_dev = NULL;
    if (hipMalloc((void**)&rgb_dev, (size_t)H * W * 3 * sizeof(float)) != hipSuccess) return 4;
    CHECK(r2l_render(ctx, pose, 0, 1, 0, H, rgb_dev, NULL));
    float* rgb = (float*)malloc((size_t)H * W * 3 * sizeof(float));
    if (hipMemcpy(rgb, rgb_dev, (size_t)H * W * 3 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 4;
    FILE* f = fopen(argv[3], "wb");
    fwrite(rgb, sizeof(float), (size_t)H * W * 3, f);
    fclose(f);
    double s = 0;
    for (size_t i = 0; i < (size_t)H * W * 3; ++i) s += rgb[i];
    printf("rendered %dx%d, n_block=%d, precision=%d, mean rgb %.6f, %lld FLOP/ray\n", H, W, n_block, prec,
           s / ((double)H * W * 3), r2l_flops_per_ray(ctx));
    r2l_destroy(ctx);
    (void)hipFree(rgb_dev);
    return 0;
}
