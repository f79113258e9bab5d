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
    // compact-intermediate side buffers (Pass1::layer_input_c), scratch of one batch like T:
    cplx *pcol;     // [launch slot][N]     P(ky) of texel column id.x = 0, in pass-2 lane order (Pass2::pcol_index)
    cplx *rrow;     // [launch slot][N x'][4] row transforms Q1..Q3 of texel row id.y = 0 (entry 0 unused)
    const cplx *tw_split; // split plan (N = 2048): [twiddle table of the N/2 plan][W_N^k, k = 0 .. N/2 - 1]; nullptr otherwise
    const cplx *tw_half;  // half table (N = 2048, the compact pass 2: plan_twh_total(N) entries, ow_device.h "HALF TABLE"); nullptr otherwise
    uint32_t *status; // device status word (page-locked host memory, mapped): kernels OR kStatus* bits into it
};

// optional events bound to a launch's own dispatch packet (begin / end of the kernel itself)
struct LaunchTiming {
    hipEvent_t start = nullptr, stop = nullptr;
};

// consumer-side sampling (ow_consumer.hip); SurfaceSample is layout-identical to ow_surface_sample in include/ocean_waves.h
struct SurfaceScales {
    float s[8][4];  // map_scales[i] = (1/tile_length.x, 1/tile_length.y, displacement_scale, normal_scale), water.gd:105-109
};
struct SurfaceSample {
    float displacement[3];
    float gradient[2];
    float gradient_scaled[2];
    float foam;
    float normal_factor, foam_factor, scale_factor;
    int32_t spray_active;
    float gradient_fragment[2];
    float foam_fragment;
    float reserved;
};
hipError_t launch_sample_surface(int n, int cascades, const DeviceBuffers &buf, const float *xz_dev, int count,
                                 const SurfaceScales &scales, SurfaceSample *out_dev, hipStream_t s);

bool supported_map_size(int n);
int kernel_family(int n, int slots, int mode);  // 1 standard, 2 layer-parallel, 3 compact: what launch_pass1/2 will use
hipError_t launch_spectrum(int n, int cascade, const SpectrumPC &pc, const DeviceBuffers &buf, hipStream_t s);
hipError_t launch_pass1(int n, int slots, int mode, const FrameArgs &args, const DeviceBuffers &buf, hipStream_t s,
                        const LaunchTiming &lt = LaunchTiming{});  // mode: 0 auto, 1 standard, 2 layer-parallel, 3 compact
hipError_t launch_pass2(int n, int slots, int mode, const FrameArgs &args, const DeviceBuffers &buf, hipStream_t s,
                        const LaunchTiming &lt = LaunchTiming{});

// Tick groups for ow_run on small batches (ow_frame_kernels.h k_tick_group_c_lp): pass 2 of g.d2 consecutive ticks (scratch slots
// g.tbase2[j] + i) and / or pass 1 of g.d1 later ticks (times g.time1[j][i], scratch slots g.tbase1[j] + i) in one launch; n2 / n1 are
// filled in by the launcher.
bool tick_groups_supported(int n);
bool tick_pairs_supported(int n);
int tick_group_pipe_blocks(int n, int slots);  // pass-2 blocks of the pipelined form (0: not available at this map size)
hipError_t launch_tick_group(int n, const FrameArgs &args, const TickGroupArgs &g, const DeviceBuffers &buf, hipStream_t s,
                             const LaunchTiming &lt = LaunchTiming{}, hipStream_t side = nullptr);
// TWO CHAINS (round 6).  A tick-pair launch of four 1024^2 cascades on either side is exactly two generations of blocks, and the kernel boundary between
// two such launches costs a tenth of them: the chip drains (the last 8 us run below half occupancy) and fills again.  Cascades are independent, so the
// launch can go out as two launches of TWO cascades each on two streams -- the first halves of both sides on `s`, the second halves on `side` -- each a
// chain of its own (a half's pass 2 needs nothing but that half's pass 1 of the launch before): one chain's drain runs under the other's body.
// Same kernel, same items, same bits.  1024^2 x 4: 52.1 -> 48.0 us per tick on one box (scripts/two_ctx.py; everything else that was tried -- halves of
// other sizes, four chains, 2048^2 -- loses: a half must still fill the chip, and the two chains together must fit the Infinity Cache).
// tick_pair_splits: would launch_tick_group split this launch when given a side stream?  The caller orders the side stream (fork / join) around it.
bool tick_pair_splits(int n, const TickGroupArgs &g);

}  // namespace ow
