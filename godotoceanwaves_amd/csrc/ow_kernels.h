// ow_kernels.h -- internal launcher interface between the host runtime and the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "ow_device.h"

namespace ow {

struct DeviceBuffers {
    cplx *h0;       // [layers][N][N] complex h0(k): first half of the `spectrum` texel (wave_generator.gd:31); the second half,
                    // conj(h0(-k)), is the mirrored texel of the same plane (Pass1::load_modulate)
    float *omega;   // [layers][N][N]          FP32 dispersion plane
    cplx *T;        // [launch slot][4 packed layers][N/16 y/16][N x'][16 y%16] complex: transposed intermediate (scratch between the
                    // two passes of ONE batch: indexed by the slot inside the launch, not by cascade, so every batch reuses it)
    u16x4 *disp;    // [layers][N][N] RGBA16F
    u16x4 *norm;    // [layers][N][N] RGBA16F (foam in .a)
    uint16_t *foam; // [layers][N x'][N/16 t][16 o] FP16: private copy of normal.a in pass-2 lane order (Pass2::foam_index)
    float *f32;     // [layers][N][N][8] or nullptr
    const cplx *tw; // twiddle table (plan_tw_total(N) entries)
};

// optional events bound to a launch's own dispatch packet (begin / end of the kernel itself)
struct LaunchTiming {
    hipEvent_t start = nullptr, stop = nullptr;
};

bool supported_map_size(int n);
hipError_t launch_spectrum(int n, int cascade, const SpectrumPC &pc, const DeviceBuffers &buf, hipStream_t s);
hipError_t launch_pass1(int n, int slots, int mode, const FrameArgs &args, const DeviceBuffers &buf, hipStream_t s,
                        const LaunchTiming &lt = LaunchTiming{});  // mode: 0 auto, 1 standard, 2 layer-parallel
hipError_t launch_empty(hipStream_t s);  // one idle wave: calibrates event-bracket overhead
hipError_t launch_pass2(int n, int slots, int mode, const FrameArgs &args, const DeviceBuffers &buf, hipStream_t s,
                        const LaunchTiming &lt = LaunchTiming{});

}  // namespace ow
