// cvtcheck.hip -- developer check (not part of the product): v_cvt_pk_f16_f32 (ow_device.h f2h2) against v_cvt_f16_f32 (f2h) over ALL 2^32
// FP32 bit patterns, in either operand slot, plus the bit-identity of the packed complex multiply (cmul with neg_lo) and of the packed
// sin/cos chain (expi_phase) with their scalar forms on 2^28 pseudo-random operands.  Prints the number of mismatches (0 expected).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I godotoceanwaves_amd/csrc tools/cvtcheck.hip -o tools/cvtcheck
#include <hip/hip_runtime.h>

#include <cstdio>

#include "ow_device.h"
using namespace ow;

__global__ void k_cvt(unsigned long long *bad) {
    const uint32_t base = (blockIdx.x * 256u + threadIdx.x) * 4096u;
    unsigned n = 0;
    for (uint32_t i = 0; i < 4096u; ++i) {
        const uint32_t bits = base + i, other = bits * 2654435761u + 12345u;
        const float a = __builtin_bit_cast(float, bits), b = __builtin_bit_cast(float, other);
        const uint32_t want_ab = (uint32_t)f2h(a) | ((uint32_t)f2h(b) << 16), want_ba = (uint32_t)f2h(b) | ((uint32_t)f2h(a) << 16);
        n += (f2h2(a, b) != want_ab) + (f2h2(b, a) != want_ba);
    }
    if (n) atomicAdd(bad, (unsigned long long)n);
}
__device__ __forceinline__ uint32_t rnd(uint32_t &s) {
    s ^= s << 13, s ^= s >> 17, s ^= s << 5;
    return s;
}
__global__ void k_cmul(unsigned long long *bad) {
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 747796405u + 2891336453u;
    unsigned n = 0;
    for (int i = 0; i < 1024; ++i) {
        float v[4];
        for (float &x : v) {  // finite values over a wide range of exponents (2^-20 .. 2^20), both signs
            const uint32_t r = rnd(s);
            x = __builtin_bit_cast(float, (r & 0x807FFFFFu) | ((107u + (r >> 23) % 41u) << 23));
        }
        const cplx a = cplx{v[0], v[1]}, b = cplx{v[2], v[3]};
        const cplx got = cmul(a, b), want = __builtin_elementwise_fma(a.yy * cplx{-1.0f, 1.0f}, b.yx, a.xx * b);
        const float gx = got.x, gy = got.y, wx = want.x, wy = want.y;
        n += (__builtin_bit_cast(uint32_t, gx) != __builtin_bit_cast(uint32_t, wx)) + (__builtin_bit_cast(uint32_t, gy) != __builtin_bit_cast(uint32_t, wy));
        // phases up to 2.5e4 rad
        const float ph = v[0] * (1.0f / 1048576.0f) * 2.5e4f;
        float sn, cs;
        sincos_phase(ph, sn, cs);
        const cplx m = expi_phase(ph);
        const float mx = m.x, my = m.y;
        n += (__builtin_bit_cast(uint32_t, mx) != __builtin_bit_cast(uint32_t, cs)) + (__builtin_bit_cast(uint32_t, my) != __builtin_bit_cast(uint32_t, sn));
    }
    if (n) atomicAdd(bad, (unsigned long long)n);
}
int main() {
    unsigned long long *bad, h[2] = {0, 0};
    if (hipMalloc(&bad, 16) != hipSuccess || hipMemset(bad, 0, 16) != hipSuccess) return 2;
    hipLaunchKernelGGL(k_cvt, dim3(4096), dim3(256), 0, 0, bad);        // 4096 * 256 * 4096 = 2^32 patterns
    hipLaunchKernelGGL(k_cmul, dim3(1024), dim3(256), 0, 0, bad + 1);   // 2^28 operand sets
    if (hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    printf("v_cvt_pk_f16_f32 vs v_cvt_f16_f32, all 2^32 inputs, both slots: %llu mismatches\n", h[0]);
    printf("cmul (neg_lo) and expi_phase vs their scalar forms, 2^28 operand sets: %llu mismatches\n", h[1]);
    return (h[0] || h[1]) ? 1 : 0;
}
