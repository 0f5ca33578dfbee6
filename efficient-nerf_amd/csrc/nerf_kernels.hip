// placeholder translation unit (teacher kernels land here)
#include "nerf_kernels.h"
