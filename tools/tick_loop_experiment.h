// tick_loop_experiment.h -- developer experiment (NOT part of the product; included by tools/kbench_small.hip only).
//
// A persistent cooperative kernel that runs up to 16 consecutive ticks of a small batch (both passes of every tick, a grid-wide
// rendezvous instead of kernel boundaries, double-buffered scratch intermediate), built from the same item bodies as the
// stand-alone layer-parallel compact kernels.  MEASURED AND REJECTED (profiles/r02_tick_loop_rejected.txt): 256^2 x 4 18.8 us per
// tick against 15.3 as one pair of launches per tick, 512^2 x 4 49.6 against 26.4, 1024^2 x 1 47.8 against 29.7.  What a kernel
// boundary costs on this part (~1.6 us of launch + ~1.5 us of kernel-argument / first-load round trips) is less than what replaces
// it inside one kernel: data that crosses XCDs has to move with sc1 (write-through / L2-miss) accesses, and the chain
// "stores acknowledged -> counter bumped -> counter polled -> loads served" is four memory-side round trips of ~2 us each.
// (A first version that fenced at agent scope instead -- L2 write-back + invalidate around the rendezvous -- took 55 us per tick.)
#pragma once
#include "ow_frame_kernels.h"

namespace ow {

constexpr uint32_t kStatusGridSyncTimeout = 2u;
constexpr int kMaxTicksPerLaunch = 16;
struct TickTimes {
    float t[kMaxTicksPerLaunch][8];  // FP32-narrowed params.time of tick k, launch slot i
};

// ===================================================================================================
// PERSISTENT TICK LOOP for small batches (the layer-parallel compact family, N <= 1024): `ticks` consecutive ticks of all
// slots in ONE cooperative launch.  A small tick is two kernels of ~6 us of which ~1.6 us each is launch / dispatch and
// ~1.3-1.9 us each the serial kernel-argument and first-load round trips before any work starts (phase stamps:
// profiles/r02_small_phase_stamps_*.txt); the loop pays them once per launch instead of once per pass and tick, keeps the
// twiddle table in LDS, and replaces the kernel boundary between the passes by a grid-wide rendezvous.
//   tick k:   phase A = pass-1 items into T[k & 1]   -> grid rendezvous ->   phase B = pass-2 items out of T[k & 1]
// The scratch intermediate (and its side buffers) is double-buffered, so ONE rendezvous per tick suffices: a block that is
// done with phase B of tick k starts phase A of tick k+1 at once, and T[k & 1] is rewritten by phase A of tick k+2 only after
// the rendezvous of tick k+1, which no block reaches before its own phase B of tick k is finished.
// Foam / maps of a row are read and written by the same block in every tick (the item -> block mapping is fixed).
// The rendezvous is a monotonically increasing counter in device memory, bumped and polled with agent-scope atomics; the data that
// crosses blocks (T, pcol, rrow) moves with sc1 accesses, so no L2 write-back / invalidate is needed around it (a first version
// that fenced at agent scope instead spent ~40 us per tick in those cache walks).  The wait is BOUNDED and a time-out is reported
// through the status word.
// ===================================================================================================

__device__ __forceinline__ bool grid_rendezvous(unsigned *counter, unsigned target, uint32_t *status) {
    // No cache maintenance: everything that crosses blocks inside the tick loop (T, pcol, rrow) is stored and loaded with sc1
    // (kAuxAgent), i.e. written through and never served from a line another XCD could have outdated.  What is left to do is to
    // wait until this wave's stores have been acknowledged before the block checks in.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spin = 0;
        // signed distance: the counter runs on across launches and may wrap.  Bounded (2^20 polls of >= 1 us: about a second)
        while ((int)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0 && spin < (1 << 20)) {
            __builtin_amdgcn_s_sleep(1);
            ++spin;
        }
        if (spin == (1 << 20)) {
            __hip_atomic_fetch_or(status, kStatusGridSyncTimeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            ok = false;
        }
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return ok;
}

template <int N, bool F32>
__global__ __launch_bounds__(plan_lp_threads(N), 4) void k_ticks_c_lp(DeviceBuffers buf, FrameArgs args, TickTimes times, int slots, int ticks,
                                                                          unsigned *counter, unsigned counter_base) {
    using TP = TickPlan<N>;
    constexpr int ROWS = plan_lp_rows(N), SUB = plan_wg_threads(N);
    static_assert(!plan_row_spans_waves(N), "the tick loop serves the small-batch sizes (N <= 1024)");
    __shared__ __attribute__((aligned(16))) cplx lds[plan_lp_lds_cplx(N)];
    cplx *tw_lds = lds;
    cplx *rows_lds = lds + plan_tw_total(N);
    RowSync<N> rs;
    NoStamps ws;
    load_twiddles<N>(tw_lds, buf.tw);  // once per launch
    const int items_a = TP::items_1(slots), items_b = TP::items_2(slots);
    const int n_full = TP::full(slots), n_upper = TP::upper(slots);
    bool alive = true, first = true;
    for (int k = 0; k < ticks; ++k) {
        const int tbase = (k & 1) * slots;  // scratch slots of this tick
        // (per phase and tick: everything derived from the lane index is recomputed, not carried in registers around the loop)
        const int tau_a = opaque((int)threadIdx.x);
        const int sub = __builtin_amdgcn_readfirstlane(tau_a / SUB), tau_sub = tau_a % SUB;  // pass-1 item of this lane inside the block
        // ---- phase A: pass 1 ----
        for (int item = blockIdx.x; item < items_a; item += gridDim.x) {
            if (!first) lds_barrier();  // the previous item's LDS reads are done
            first = false;
            int L, slot, row0;
            bool active = true;
            if (item < 2 * n_full) {  // layers 0 and 2: every 8-row group
                L = item < n_full ? 0 : 2;
                const int group = (item < n_full ? item : item - n_full) * TP::Q + sub;
                slot = group / TP::GPS;
                row0 = (group % TP::GPS) * kWgRows;
            } else if (item < 2 * n_full + n_upper) {  // layer 1: the upper half of the rows
                L = 1;
                const int group = (item - 2 * n_full) * TP::Q + sub;
                slot = group / (TP::GPS / 2);
                row0 = N / 2 + (group % (TP::GPS / 2)) * kWgRows;
            } else {  // the three extra transforms of texel row 0, one (slot, Q) pair per sub-block
                const int r = (item - 2 * n_full - n_upper) * TP::Q + sub;
                active = r < slots * 3;
                slot = active ? r / 3 : 0;
                L = 3 + (active ? r % 3 : 0);
                row0 = 0;
            }
            if (active) {  // (sub-block uniform; the row-0 path has no block barrier, the layer paths are block uniform)
                const CascadeFrame cf = args.c[slot];
                pass1c_lp_item<N, kAuxAgent>(buf, cf, times.t[k][slot], tbase + slot, row0, L, tau_sub, tw_lds,
                                               rows_lds + sub * kWgRows * plan_region_cplx(N), rs, [] {}, ws);
            }
        }
        if (alive) alive = grid_rendezvous(counter, counter_base + (unsigned)(k + 1) * gridDim.x, buf.status);
        // ---- phase B: pass 2 ----
        const int tau = opaque((int)threadIdx.x);
        for (int item = blockIdx.x; item < items_b; item += gridDim.x) {
            if (!first) lds_barrier();
            first = false;
            const int slot = item / (N / ROWS), row0 = (item % (N / ROWS)) * ROWS;
            CascadeFrame cf = args.c[slot];
            cplx foam_bits;
            pass2c_lp_item<N, F32, kAuxAgent, kAuxDefault>(buf, cf, tbase + slot, row0, tau, tw_lds, rows_lds, rs, [] {}, ws, foam_bits);
        }
    }
}

}  // namespace ow
