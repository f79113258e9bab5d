// sinbench.hip -- accuracy of the hardware v_sin_f32 / v_cos_f32 (argument in revolutions) on [-0.25, 0.25], the range a
// Cody-Waite reduction by pi would hand it.  hipcc -O3 --offload-arch=gfx950 tools/sinbench.hip -o tools/build/sinbench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float *x, float *s, float *c, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { s[i] = __builtin_amdgcn_sinf(x[i]); c[i] = __builtin_amdgcn_cosf(x[i]); }
}
int main() {
    const int n = 1 << 22;
    std::vector<float> x(n), s(n), c(n);
    for (int i = 0; i < n; ++i) x[i] = -0.25f + 0.5f * (float)i / (float)(n - 1);
    float *dx, *ds, *dc;
    hipMalloc(&dx, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, ds, dc, n);
    hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
    double es = 0, ec = 0, xs = 0, xc = 0;
    for (int i = 0; i < n; ++i) {
        const double a = 2.0 * M_PI * (double)x[i];
        const double e1 = fabs((double)s[i] - sin(a)), e2 = fabs((double)c[i] - cos(a));
        if (e1 > es) { es = e1; xs = x[i]; }
        if (e2 > ec) { ec = e2; xc = x[i]; }
    }
    printf("v_sin_f32 max abs err %.3e at %.6f rev ; v_cos_f32 max abs err %.3e at %.6f rev\n", es, xs, ec, xc);
    return 0;
}
