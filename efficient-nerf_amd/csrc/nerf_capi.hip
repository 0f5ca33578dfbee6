// C-ABI host side for the NeRF teacher (include/r2l_hip.h).  WORK IN PROGRESS: entry
// points fail loudly until the kernels land.
#include <hip/hip_runtime.h>
#include "../../include/r2l_hip.h"
#include "r2l_host_util.h"
#include "nerf_kernels.h"

#define NI(name) return r2l_set_error(R2L_EINVAL, name ": not implemented in this build")
int nerf_create(nerf_ctx** out, int, int, double, float, float, int, int, int, int, int, int) { if (out) *out = nullptr; NI("nerf_create"); }
void nerf_destroy(nerf_ctx*) {}
int nerf_load_weights(nerf_ctx*, int, const float* const*, int) { NI("nerf_load_weights"); }
int nerf_render(nerf_ctx*, const float*, int, int, float*, float*, float*, float*, void*) { NI("nerf_render"); }
int nerf_render_rays(nerf_ctx*, const float*, const float*, int, float*, float*, float*, float*, void*) { NI("nerf_render_rays"); }
int nerf_last_extras(nerf_ctx*, const float**, const float**, const float**, const float**) { NI("nerf_last_extras"); }
int nerf_raw2outputs(const float*, const float*, const float*, int, int, int, float*, float*, float*, float*, float*, void*) { NI("nerf_raw2outputs"); }
int nerf_sample_pdf(const float*, const float*, int, int, int, float*, void*) { NI("nerf_sample_pdf"); }
int nerf_merge_sorted(const float*, int, const float*, int, int, float*, void*) { NI("nerf_merge_sorted"); }
