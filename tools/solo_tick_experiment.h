// solo_tick_experiment.h -- developer experiment (NOT part of the product; included by tools/kbench_solo.hip only).
//
// VERDICT r3 task 6: a lone small tick (what ow_update_all / ow_process callers get: no look-ahead across ticks) in ONE launch instead
// of two.  Blocks [0, slots * n1) are pass-1 items, the rest pass-2 items that wait -- per cascade, not grid-wide -- until all n1
// pass-1 items of THEIR cascade have arrived at a counter; nothing else orders them, and their own preamble (kernel arguments, twiddle
// table, address arithmetic: 2.8 us of a stand-alone pass 2) runs under pass 1.  Two forms of the hand-off:
//   ANY PLACEMENT  the intermediate crosses with sc1 stores (write-through) and sc1 loads: valid wherever the blocks run
//                  (MI355X_MICROARCH.md "inter-workgroup visibility": per-XCD L2s are not coherent with each other);
//   XCD-LOCAL      block b works on cascade b % 8, i.e. -- by the dispatcher's observed round-robin -- a cascade's two passes share one XCD
//                  and its L2: plain stores (lines stay in that L2), sc1 loads (skip the CU's L1).  Correct only while the placement
//                  holds, which the platform does not promise: every block checks its XCC id and counts violations.
#pragma once
#include "ow_frame_kernels.h"

namespace ow {

struct SoloArgs {
    unsigned target[8];  // counter value at which cascade slot i's pass 1 is complete (the counters run on across launches)
    int slots;
};

template <int N, bool F32, bool LOCAL>
__global__ __launch_bounds__(plan_lp_threads(N), 4) void k_tick_solo_c_lp(DeviceBuffers buf, FrameArgs args, SoloArgs sa, unsigned *counters, unsigned *misplaced) {
    using TP = TickPlan<N>;
    constexpr int ROWS = plan_lp_rows(N), SUB = plan_wg_threads(N);
    constexpr int kStoreAux = LOCAL ? kAuxDefault : kAuxAgent, kLoadAux = kAuxAgent;
    static_assert(!plan_row_spans_waves(N), "small-batch sizes only");
    __shared__ __attribute__((aligned(16))) cplx lds[plan_lp_lds_cplx(N)];
    cplx *tw_lds = lds;
    cplx *rows_lds = lds + plan_tw_total(N);
    RowSync<N> rs;
    NoStamps ws;
    TwPrefetch<N> twp;
    tw_fetch<N>(twp, buf.tw);
    const int n1 = TP::items_1(1), n2 = TP::items_2(1);
    int slot, j;
    if constexpr (LOCAL) {  // cascade = XCD: blocks b, b + 8, b + 16, ... of one cascade
        slot = blockIdx.x & 7;
        j = blockIdx.x >> 3;
        if (slot >= sa.slots) return;
        if (threadIdx.x == 0 && xcc_id() != (unsigned)slot) atomicAdd(misplaced, 1u);
    } else {  // all pass-1 blocks first (they are dispatched before any block that waits), cascade-major inside each kind
        const int b = blockIdx.x;
        if (b < sa.slots * n1) {
            slot = b / n1;
            j = b % n1;
        } else {
            slot = (b - sa.slots * n1) / n2;
            j = n1 + (b - sa.slots * n1) % n2;
        }
    }
    const CascadeFrame cf = args.c[slot];
    fetch_arguments(buf, cf);
    if (j < n1) {  // ---- a pass-1 item: Q side-by-side (8 rows, layer) sub-items ----
        const int tau = threadIdx.x, sub = __builtin_amdgcn_readfirstlane(tau / SUB), tau_sub = tau % SUB;
        int L, s0, row0;
        const bool active = TP::decode(j, sub, 1, L, s0, row0);
        if (active) {
            pass1c_lp_item<N, kStoreAux>(buf, cf, cf.time, slot, row0, L, tau_sub, tw_lds, rows_lds + sub * kWgRows * plan_region_cplx(N), rs,
                                         [&] { tw_commit<N>(twp, tw_lds); }, ws);
        } else {
            tw_commit<N>(twp, tw_lds);
        }
        // publish: every writing wave drains its own stores, then ONE lane arrives
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(counters + slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // ---- a pass-2 item: everything that does not depend on pass 1 first, then the wait, then the loads ----
    const int row0 = (j - n1) * ROWS;
    tw_commit<N>(twp, tw_lds);
    if (threadIdx.x == 0) {
        int spin = 0;
        while ((int)(__hip_atomic_load(counters + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - sa.target[slot]) < 0 && spin < (1 << 20)) {
            __builtin_amdgcn_s_sleep(1);
            ++spin;
        }
        if (spin == (1 << 20)) __hip_atomic_fetch_or(buf.status, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    cplx foam_bits;
    pass2c_lp_item<N, F32, kLoadAux, kAuxDefault>(buf, cf, slot, row0, (int)threadIdx.x, tw_lds, rows_lds, rs, [] {}, ws, foam_bits);
}

}  // namespace ow
