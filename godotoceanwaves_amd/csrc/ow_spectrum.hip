// ow_spectrum.hip -- one-time (per parameter change) spectrum generation for gfx950.
// Replaces the spectrum_compute.glsl dispatch (wave_generator.gd:68-72) and additionally writes the
// FP32 dispersion plane omega(k) that spectrum_modulate.glsl:65 would otherwise recompute every frame.
// Built with -ffp-contract=off: omega must be bit-identical to the oracle's.
#include "ow_kernels.h"

namespace ow {

__global__ __launch_bounds__(256) void k_spectrum(int n, int cascade, SpectrumPC pc, DeviceBuffers buf) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= n || y >= n) return;
    const size_t idx = ((size_t)cascade * n + y) * n + x;
    buf.h0[idx] = spectrum_amplitude_fast(x, y, n, pc);  // the texel's .zw is the mirrored texel's .xy, conjugated: not stored
    buf.omega[idx] = omega_texel(x, y, n, pc.tile_x, pc.tile_y, pc.depth);
}

hipError_t launch_spectrum(int n, int cascade, const SpectrumPC &pc, const DeviceBuffers &buf, hipStream_t s) {
    hipLaunchKernelGGL(k_spectrum, dim3(n / 64, n / 4), dim3(256), 0, s, n, cascade, pc, buf);
    return hipGetLastError();
}

}  // namespace ow
