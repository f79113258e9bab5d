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
    // intermediate: layer-parallel from 256^2 x 1 (64 waves; on a single tick it ties with the four-layer pair, 11.8 vs 11.6 us, but
    // ow_run's tick groups exist for this family only: 9.0 vs 11.6 us per tick) up to 1536 waves
    // (round 2, scripts/pairs_crossover.py, us per tick of ow_run -- layer-parallel compact as tick pairs / one launch per pass
    // against k_pass1c + k_pass2c: 512^2 x 5 27.1 / 30.9 vs 35.1; 512^2 x 6 28.6 / 33.8 vs 36.6; 512^2 x 8 41.6 / 42.8 vs 37.6;
    // 1024^2 x 2 41.3 / 45.1 vs 39.7).
    if (has_lp_compact && waves >= 64 && waves <= 1536) return 4;
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
            launch(k_pass1c_split<N, OW_SPLIT_P1_ROWS>, dim3(blocks * (kWgRows / OW_SPLIT_P1_ROWS)), dim3(SplitGeo<N, OW_SPLIT_P1_ROWS>::kThreads), s, lt, buf, args, (Stamp *)nullptr);
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
            // The two output maps are stored non-temporally: they are write-only streams of 16 B/texel that nothing on the device
            // reads back soon, and left to the default policy they push the spectra and the intermediate out of the 256 MiB Infinity
            // Cache as soon as a tick's working set exceeds it (round 2, same box: 2048^2 x 4 303 -> 272 us per tick, 1024^2 x 8
            // 124 -> 115, 512^2 x 8 38.0 -> 35.1; 1024^2 x 4, which fits, 57.1 -> 57.2).  nt on the T loads -- T is dead after
            // this pass -- loses (1024^2 x 4: 61.1), sc1 on the outputs changes nothing, nt on the T stores of pass 1 loses (2048^2 x 4: 307).
            if (buf.f32) launch(k_pass2c<N, true, kAuxDefault, kAuxNT>, dim3(blocks), dim3(plan_wg_threads(N)), s, lt, buf, args);
            else launch(k_pass2c<N, false, kAuxDefault, kAuxNT>, dim3(blocks), dim3(plan_wg_threads(N)), s, lt, buf, args);
            return hipGetLastError();
        }
    }
    if (buf.f32) launch(k_pass2<N, true>, dim3(blocks), dim3(plan_wg_threads(N)), s, lt, buf, args, DebugArgs{});
    else launch(k_pass2<N, false>, dim3(blocks), dim3(plan_wg_threads(N)), s, lt, buf, args, DebugArgs{});
    return hipGetLastError();
}

// ---- tick pairs on the compact family (k_tick_pair_c): k_pass2c's blocks of one batch, k_pass1c's of the next, in one launch ----
static int chain_slots(int n) {  // a side of a tick-pair launch that two chains share: four 1024^2 cascades (two each), eight 512^2 cascades (four each)
    return n == 1024 ? 4 : n == 512 ? 8 : 0;
}
bool tick_pair_splits(int n, const TickGroupArgs &g) {
    const int cs = chain_slots(n);
    return cs > 0 && g.pair_compact && (g.slots2 == cs || g.slots2 == 0) && (g.slots1 == cs || g.slots1 == 0) && g.slots2 + g.slots1 > 0;
}
template <int N, bool F32>
static hipError_t launch_pair_n(const FrameArgs &args, TickGroupArgs g, const DeviceBuffers &buf, hipStream_t s, const LaunchTiming &lt, hipStream_t side = nullptr) {
    if (g.slots2 < 0 || g.slots1 < 0 || g.first2 < 0 || g.first1 < 0 || g.first2 + g.slots2 > kMaxCascades || g.first1 + g.slots1 > kMaxCascades)
        return hipErrorInvalidValue;
    if (side && !lt.start && tick_pair_splits(N, g)) {  // two chains: launch slots [first, first + 2) of both sides on s, [first + 2, first + 4) on side
        TickGroupArgs h = g;
        h.slots2 = g.slots2 / 2, h.slots1 = g.slots1 / 2;
        const hipError_t e = launch_pair_n<N, F32>(args, h, buf, s, lt);
        if (e != hipSuccess) return e;
        h.first2 = g.first2 + h.slots2, h.first1 = g.first1 + h.slots1;
        h.tbase2[0] = g.tbase2[0] + h.slots2, h.tbase1[0] = g.tbase1[0] + h.slots1;  // (scratch slot = tbase + index inside the batch)
        return launch_pair_n<N, F32>(args, h, buf, side, lt);
    }
    constexpr int per2 = plan_split(N) ? N / PairSplitGeo<N>::kCols : N / kWgRows;  // blocks per cascade: 4 columns / 4 rows at 2048 (8-wave blocks of
    constexpr int per1 = plan_split(N) ? N / 4 : N / kWgRows;                       // both kinds), 8 columns / 8 rows below
    g.n2 = g.slots2 * per2;
    g.n1 = g.slots1 * per1;
    if (g.n2 + g.n1 < 1) return hipErrorInvalidValue;
    PairArgs pa;
    for (int i = 0; i < kMaxCascades; ++i) {
        const CascadeFrame &cf = args.c[i];
        pa.c[i] = PairFrame{cf.tile_x, cf.tile_y, cf.whitecap, cf.foam_grow_rate, cf.foam_decay, cf.cascade};
        pa.time1[i] = g.time1[0][i];
    }
    pa.tbase2 = g.tbase2[0], pa.tbase1 = g.tbase1[0], pa.n2 = g.n2, pa.n1 = g.n1, pa.first2 = g.first2, pa.first1 = g.first1;
    pa.fault = args.c[0].fault, pa.pad = 0;
    if constexpr (plan_split(N)) launch(k_tick_pair_c_split<N, F32>, dim3(g.n2 + g.n1), dim3(PairSplitGeo<N>::kThreads), s, lt, buf, pa, (Stamp *)nullptr);
    else launch(k_tick_pair_c<N, F32>, dim3(g.n2 + g.n1), dim3(plan_wg_threads(N)), s, lt, buf, pa);
    return hipGetLastError();
}
// ---- tick groups (k_tick_group_c_lp): pass 2 of d2 ticks and pass 1 of d1 later ticks in one launch ----
template <int N, bool F32>
static hipError_t launch_group_n(const FrameArgs &args, TickGroupArgs g, const DeviceBuffers &buf, hipStream_t s, const LaunchTiming &lt, hipStream_t side) {
    using TP = TickPlan<N>;
    if (g.pair_compact) return launch_pair_n<N, F32>(args, g, buf, s, lt, side);
    if (plan_lp_rows(N) < 2) g.p2_pipe = 0;
    if (g.slots < 1 || g.first1 < 0 || g.step1 < 0 || g.first1 + (g.d1 > 0 ? g.d1 - 1 : 0) * g.step1 + g.slots > kMaxCascades) return hipErrorInvalidValue;
    g.n2 = g.d2 > 0 ? (g.p2_pipe ? TP::items_2_pipe(g.slots) : TP::items_2(g.slots)) : 0;
    g.n1 = g.p1_compact ? TP::items_1_compact(g.slots) : TP::items_1(g.slots);
    const int blocks = g.n2 + g.d1 * g.n1;
    if (blocks < 1) return hipErrorInvalidValue;
    if (g.p2_pipe) launch(k_tick_group_c_lp<N, F32, false, true>, dim3(blocks), dim3(plan_lp_threads(N)), s, lt, buf, args, g, (Stamp *)nullptr);
    else launch(k_tick_group_c_lp<N, F32>, dim3(blocks), dim3(plan_lp_threads(N)), s, lt, buf, args, g, (Stamp *)nullptr);
    return hipGetLastError();
}
bool tick_groups_supported(int n) { return n == 256 || n == 512 || n == 1024; }
bool tick_pairs_supported(int n) { return tick_groups_supported(n) || n == 2048; }  // (2048: k_tick_pair_c_split, one cascade per batch)
int tick_group_pipe_blocks(int n, int slots) {
    return n == 256 ? TickPlan<256>::items_2_pipe(slots) : n == 512 ? TickPlan<512>::items_2_pipe(slots) : n == 1024 ? TickPlan<1024>::items_2_pipe(slots) : 0;
}
hipError_t launch_tick_group(int n, const FrameArgs &args, const TickGroupArgs &g, const DeviceBuffers &buf, hipStream_t s, const LaunchTiming &lt, hipStream_t side) {
    if (g.d2 < 0 || g.d1 < 0 || g.d2 > kMaxTickGroup || g.d1 > kMaxTickGroup) return hipErrorInvalidValue;
#define OW_GROUP(NN) \
    case NN: return buf.f32 ? launch_group_n<NN, true>(args, g, buf, s, lt, side) : launch_group_n<NN, false>(args, g, buf, s, lt, side);
    switch (n) {
        OW_GROUP(256)
        OW_GROUP(512)
        OW_GROUP(1024)
        case 2048:  // tick pairs only (a row spans two waves: no tick groups at this size)
            if (!g.pair_compact) return hipErrorInvalidValue;
            return buf.f32 ? launch_pair_n<2048, true>(args, g, buf, s, lt) : launch_pair_n<2048, false>(args, g, buf, s, lt);
    }
#undef OW_GROUP
    return hipErrorInvalidValue;
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
