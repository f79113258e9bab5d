// ow_frame.hip -- product instantiations + launchers of the two per-frame kernels (gfx950 / MI355X).
#include <hip/hip_ext.h>

#include "ow_frame_kernels.h"

namespace ow {

// Launch with optional start/stop events bound to the dispatch packet itself (hipExtLaunchKernelGGL): their elapsed
// time is the kernel's own begin -> end, the figure a rocprofv3 kernel trace reports.
template <class K, class... A>
static void launch(K kernel, dim3 grid, dim3 block, hipStream_t s, const LaunchTiming &lt, A... args) {
    if (lt.start) hipExtLaunchKernelGGL(kernel, grid, block, 0, s, lt.start, lt.stop, 0, args...);
    else hipLaunchKernelGGL(kernel, grid, block, 0, s, args...);
}

// family the runtime picks for batches too large for the layer-parallel kernels, where the compact kernels exist
constexpr int kAutoLargeFamily = 3;  // measured (scripts/mode_bench.py): 1024^2 x 4 65.1 vs 71.2 us, 1024^2 x 2 42.0 vs 44.5, 2048^2 x 1 71.7 vs 81.5

bool supported_map_size(int n) { return n == 128 || n == 256 || n == 512 || n == 1024 || n == 2048; }

// Kernel family of a batch.  mode: 0 = choose by size, 1 = standard (four-layer intermediate in the reference's packing),
// 2 = layer-parallel, 3 = compact (two-and-a-half-layer intermediate), 4 = layer-parallel on the compact intermediate.
// Small batches take the layer-parallel kernels: up to ~1024 waves of row work the standard kernels cannot fill the chip
// and run at one wave's serial latency.
template <int N>
static int family(int slots, int mode) {
    constexpr bool has_compact = plan_T(N) >= 16, has_lp_compact = plan_T(N) >= 16;
    if (mode == 1 || mode == 2) return mode;
    if (mode == 3) return has_compact ? 3 : 1;
    if (mode == 4) return has_lp_compact ? 4 : 2;
    const long waves = (long)slots * N * plan_T(N) / 64;
    // measured crossovers (scripts/mode_bench.py).  Four-layer kernels (N = 128): layer-parallel up to 1024 waves.  Compact
    // intermediate: layer-parallel from 256 waves (256^2 x 1 loses 3 % and stays on the four-layer pair) up to 1280
    // (512^2 x 5: 32.8 vs 35.9 us; 512^2 x 6: 36.1 vs 35.2; 1024^2 x 2: 48.3 vs 39.3).
    if (has_lp_compact && waves >= 256 && waves <= 1280) return 4;
    if (waves <= 1024) return 2;
    return has_compact ? kAutoLargeFamily : 1;
}
template <int N>
static hipError_t launch1(int slots, int mode, const FrameArgs &args, const DeviceBuffers &buf, hipStream_t s, const LaunchTiming &lt) {
    const int blocks = slots * (N / kWgRows), fam = family<N>(slots, mode);
    if (fam == 2) {
        launch(k_pass1_lp<N>, dim3(blocks, kLayers), dim3(plan_wg_threads(N)), s, lt, buf, args);
        return hipGetLastError();
    }
    if constexpr (plan_T(N) >= 16) {
        if (fam == 4) {
            launch(k_pass1c_lp<N>, dim3(blocks, 6), dim3(plan_wg_threads(N)), s, lt, buf, args, (Stamp *)nullptr);
            return hipGetLastError();
        }
    }
    if constexpr (plan_split(N)) {
        if (fam == 3) {  // a row spans two waves: the split plan (one wave per parity, no rendezvous in pass 1)
            launch(k_pass1c_split<N>, dim3(blocks * (kWgRows / OW_SPLIT_P1_ROWS)), dim3(SplitGeo<N, OW_SPLIT_P1_ROWS>::kThreads), s, lt, buf, args);
            return hipGetLastError();
        }
    }
    if constexpr (plan_T(N) >= 16) {
        if (fam == 3) {
            launch(k_pass1c<N>, dim3(blocks), dim3(plan_wg_threads(N)), s, lt, buf, args, (Stamp *)nullptr);
            return hipGetLastError();
        }
    }
    launch(k_pass1<N>, dim3(blocks), dim3(plan_wg_threads(N)), s, lt, buf, args, DebugArgs{});
    return hipGetLastError();
}
template <int N>
static hipError_t launch2(int slots, int mode, const FrameArgs &args, const DeviceBuffers &buf, hipStream_t s, const LaunchTiming &lt) {
    const int fam = family<N>(slots, mode);
    if (fam == 2 || fam == 4) {
        const int lp_blocks = slots * (N / plan_lp_rows(N));
        if constexpr (plan_T(N) >= 16) {
            if (fam == 4) {
                if (buf.f32) launch(k_pass2c_lp<N, true>, dim3(lp_blocks), dim3(plan_lp_threads(N)), s, lt, buf, args, (Stamp *)nullptr);
                else launch(k_pass2c_lp<N, false>, dim3(lp_blocks), dim3(plan_lp_threads(N)), s, lt, buf, args, (Stamp *)nullptr);
                return hipGetLastError();
            }
        }
        if (buf.f32) launch(k_pass2_lp<N, true>, dim3(lp_blocks), dim3(plan_lp_threads(N)), s, lt, buf, args);
        else launch(k_pass2_lp<N, false>, dim3(lp_blocks), dim3(plan_lp_threads(N)), s, lt, buf, args);
        return hipGetLastError();
    }
    const int blocks = slots * (N / kWgRows);
    if constexpr (plan_T(N) >= 16) {
        if (fam == 3) {
            if (buf.f32) launch(k_pass2c<N, true>, dim3(blocks), dim3(plan_wg_threads(N)), s, lt, buf, args);
            else launch(k_pass2c<N, false>, dim3(blocks), dim3(plan_wg_threads(N)), s, lt, buf, args);
            return hipGetLastError();
        }
    }
    if (buf.f32) launch(k_pass2<N, true>, dim3(blocks), dim3(plan_wg_threads(N)), s, lt, buf, args, DebugArgs{});
    else launch(k_pass2<N, false>, dim3(blocks), dim3(plan_wg_threads(N)), s, lt, buf, args, DebugArgs{});
    return hipGetLastError();
}

int kernel_family(int n, int slots, int mode) {
    switch (n) {
        case 128: return family<128>(slots, mode);
        case 256: return family<256>(slots, mode);
        case 512: return family<512>(slots, mode);
        case 1024: return family<1024>(slots, mode);
        case 2048: return family<2048>(slots, mode);
    }
    return 0;
}

hipError_t launch_pass1(int n, int slots, int mode, const FrameArgs &args, const DeviceBuffers &buf, hipStream_t s, const LaunchTiming &lt) {
    switch (n) {
        case 128: return launch1<128>(slots, mode, args, buf, s, lt);
        case 256: return launch1<256>(slots, mode, args, buf, s, lt);
        case 512: return launch1<512>(slots, mode, args, buf, s, lt);
        case 1024: return launch1<1024>(slots, mode, args, buf, s, lt);
        case 2048: return launch1<2048>(slots, mode, args, buf, s, lt);
    }
    return hipErrorInvalidValue;
}
hipError_t launch_pass2(int n, int slots, int mode, const FrameArgs &args, const DeviceBuffers &buf, hipStream_t s, const LaunchTiming &lt) {
    switch (n) {
        case 128: return launch2<128>(slots, mode, args, buf, s, lt);
        case 256: return launch2<256>(slots, mode, args, buf, s, lt);
        case 512: return launch2<512>(slots, mode, args, buf, s, lt);
        case 1024: return launch2<1024>(slots, mode, args, buf, s, lt);
        case 2048: return launch2<2048>(slots, mode, args, buf, s, lt);
    }
    return hipErrorInvalidValue;
}

}  // namespace ow
