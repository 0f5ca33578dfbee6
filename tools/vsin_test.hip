// accuracy of v_sin_f32 (sin(2*pi*x), x in revolutions) on [-0.25, 0.25] vs double
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
__global__ void k(const float* x, float* y, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = __builtin_amdgcn_sinf(x[i]);
}
int main() {
    const int n = 1 << 22;
    std::vector<float> x(n), y(n);
    for (int i = 0; i < n; ++i) x[i] = -1.0f + 2.0f * (float)i / (float)(n - 1);
    float *dx, *dy;
    hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dy, n);
    hipMemcpy(y.data(), dy, n * 4, hipMemcpyDeviceToHost);
    double worst = 0, wx = 0;
    for (int i = 0; i < n; ++i) {
        double e = fabs((double)y[i] - sin(2 * M_PI * (double)x[i]));
        if (e > worst) { worst = e; wx = x[i]; }
    }
    printf("v_sin_f32 max abs err on [-1,1]: %.3e at x=%.6f\n", worst, wx);
    return 0;
}
