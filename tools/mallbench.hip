// mallbench.hip -- does a buffer written by one kernel come back from the Infinity Cache (MALL) when the next kernel reads it?
// hipcc -O3 --offload-arch=gfx950 tools/mallbench.hip -o gpurun_out/mallbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k_write(f4 *p, size_t n, float v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = f4{v, v, v, v};
}
__global__ void k_read(const f4 *p, size_t n, float *out) {
    f4 acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
__global__ void k_copy(const f4 *a, f4 *b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main() {
    const size_t flush_bytes = 2ull << 30;
    f4 *flush, *x, *y; float *out;
    CK(hipMalloc(&flush, flush_bytes)); CK(hipMalloc(&x, 512ull << 20)); CK(hipMalloc(&y, 512ull << 20)); CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 16, block = 256;
    auto ms = [&](auto launch) { (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float t = 0; (void)hipEventElapsedTime(&t, e0, e1); return t; };
    auto do_flush = [&]() { hipLaunchKernelGGL(k_read, dim3(grid), dim3(block), 0, 0, flush, flush_bytes / 16, out); };
    // warm clocks
    for (int i = 0; i < 20; ++i) do_flush();
    (void)hipDeviceSynchronize();
    for (size_t mib : {16, 32, 64, 128, 192, 256, 384}) {
        const size_t bytes = mib << 20, n = bytes / 16;
        float cold = 0, warm_r = 0, raw = 0, wr = 0, cp_cold = 0;
        const int reps = 5;
        for (int r = 0; r < reps; ++r) {
            do_flush();
            cold += ms([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(block), 0, 0, x, n, out); });
            warm_r += ms([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(block), 0, 0, x, n, out); });
            do_flush();
            wr += ms([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(block), 0, 0, x, n, 1.0f); });
            raw += ms([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(block), 0, 0, x, n, out); });
            do_flush();
            cp_cold += ms([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(block), 0, 0, x, y, n); });
        }
        auto gbps = [&](float t, double mult) { return mult * bytes / (t / reps * 1e-3) / 1e9; };
        printf("%4zu MiB  read cold %6.0f GB/s | read again %6.0f | write %6.0f | read after write %6.0f | copy (r+w) %6.0f\n", mib,
               gbps(cold, 1), gbps(warm_r, 1), gbps(wr, 1), gbps(raw, 1), gbps(cp_cold, 2));
    }
    return 0;
}
