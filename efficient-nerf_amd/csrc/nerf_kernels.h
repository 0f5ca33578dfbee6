// NeRF-teacher kernels: launcher prototypes (filled in by nerf_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
