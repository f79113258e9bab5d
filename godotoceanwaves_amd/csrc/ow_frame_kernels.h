// ow_frame_kernels.h -- the two per-frame kernels (templates).  Included by ow_frame.hip (product) and by
// tools/kbench.hip (developer ablation benchmark: VAR bits switch parts of the kernel off).
//
//   k_pass1 : h0 + omega --(time modulate, spectrum_modulate.glsl)--> 4 packed spectra
//             --(row IFFT, fft_compute.glsl 1st dispatch)--> transposed store (transpose.glsl fused)
//   k_pass2 : row IFFT (fft_compute.glsl 2nd dispatch) --> fft_unpack.glsl fused (sign, Jacobian,
//             foam RMW, RGBA16F stores)
//
// N/16 lanes own one map row (16 points per lane) and transform its 4 packed layers one after the other,
// so a lane never holds more than one layer in flight (<= 128 VGPRs: 4 waves per SIMD, 16 rows of a
// 1024-map resident per CU).  The two LDS exchanges of a row transform stay inside the wave for N <= 1024
// (DS instructions of one wave execute in order; wave_sync() only pins the compiler's ordering).
// A workgroup is the lanes of 8 consecutive rows: they share one LDS copy of the twiddle table and, in
// pass 1, transpose each finished layer through LDS so that T is written in full 64-byte requests.
// Workgroup barriers are LDS-only (lds_barrier): they never wait for global loads or stores, so the stores
// of one layer drain underneath the next layer's butterflies.
#pragma once
#include <type_traits>

#include "ow_kernels.h"

#ifndef OW_P1_WAVES
#define OW_P1_WAVES 4
#endif
namespace ow {

// VAR bits (kbench only; the product instantiates VAR = 0):
//   1 = no global loads (synthetic data), 2 = no stores, 4 = no FFT, 8 = per-wave timestamps
struct Stamp {
    unsigned long long t[16];
    unsigned xcc, pad;
    unsigned long long rt0, rt1;  // s_memrealtime (100 MHz, one counter for the device) at the wave's start and end: the cycle counters of t[] are per CU
};
struct DebugArgs {
    Stamp *stamps;
    int never_true;  // keeps the results of the "no stores" variants alive
};

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// workgroup barrier that orders LDS traffic only: s_waitcnt lgkmcnt(0) + s_barrier, no vmcnt drain
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// Sync between the lanes that share a row's LDS region.  N <= 1024: the row is one wave, whose DS instructions execute
// in order -- nothing to wait for, only the compiler's ordering is pinned.  N = 2048: the row is a PAIR of waves; they
// rendezvous through two LDS words (each wave publishes a monotonically increasing epoch after its own LDS traffic has
// completed, and polls its partner's) instead of an s_barrier, which would put all 16 waves of the block in lockstep
// at every exchange although only pairs exchange data.
// polls of the partner's epoch word before the waiting wave starts to sleep between polls (measured, round 3: profiles/EXPERIMENTS.md)
#ifndef OW_ROWSYNC_FREE_POLLS
#define OW_ROWSYNC_FREE_POLLS 0
#endif
// (The workgroup's LDS barrier in the 8-wave blocks of the 2048^2 pair kernel instead -- 1 000 fewer instructions, no spin -- was measured in round 6
//  and is SLOWER: 59 - 62 against 56 - 60 us per cascade, profiles/r06_2048_pair_variants_rejected.txt.)
template <int N>
struct RowSync {
    typedef __attribute__((address_space(3))) int lds_int;
    volatile lds_int *mine = nullptr, *partner = nullptr;
    int epoch = 0;
    uint32_t *status = nullptr;  // device status word: a rendezvous that gives up says so there (never silently)
    bool mute = false;           // fault injection: this wave never publishes (tests only)
    bool dead = false;           // a rendezvous already timed out: no further waiting in this kernel
    // flags: two ints per pair (zeroed by init_row_sync); pair = index of the row (or row x layer) inside the block
    __device__ __forceinline__ void attach(int *flags, int pair, int wave_in_pair) {
        if constexpr (plan_row_spans_waves(N)) {
            mine = (volatile lds_int *)(flags + 2 * pair + wave_in_pair);
            partner = (volatile lds_int *)(flags + 2 * pair + (wave_in_pair ^ 1));
        }
    }
    __device__ __forceinline__ void watch(uint32_t *status_word, int fault) {
        if constexpr (plan_row_spans_waves(N)) {
            status = status_word;
            mute = (fault & kFaultRowSync) != 0 && ((threadIdx.x / 64) & 1) != 0;
        }
    }
    __device__ __forceinline__ void sync() {
        if constexpr (plan_row_spans_waves(N)) {
            ++epoch;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");  // s_waitcnt lgkmcnt(0): my LDS reads/writes are done
            if (!mute) *mine = epoch;
            // bounded: a partner that never arrives (a bug) must not hang the device.  The bound is WALL TIME, not a poll count: ~20 ms
            // of the 100 MHz constant clock (checked every 256 polls), four orders of magnitude beyond the kernel's whole duration, so
            // that a stalled partner (a page migration under caller-owned buffers, a debugger) is not mistaken for a missing one.
            // Giving up is REPORTED: the status word makes the next synchronising call fail, the results of this batch are garbage
            // (and its foam state with them: ocean_waves.h, ow_sync).  The report is a plain system-scope store of the one bit that
            // exists, followed by a system fence -- no read-modify-write, so it does not depend on PCIe atomics towards host memory.
            if (!dead) {
                int spin = 0;
                unsigned long long t0 = 0;
                bool gave_up = false;
                while (__builtin_amdgcn_readfirstlane(*partner) < epoch) {
                    if (spin >= OW_ROWSYNC_FREE_POLLS) __builtin_amdgcn_s_sleep(1);
                    if ((++spin & 255) == 0) {
                        const unsigned long long now = wall_clock64();
                        if (t0 == 0) t0 = now;
                        else if (now - t0 > kRowSyncTimeoutTicks) {
                            gave_up = true;
                            break;
                        }
                    }
                }
                if (gave_up) {
                    dead = true;
                    if ((threadIdx.x & 63) == 0) {
                        __hip_atomic_store(status, kStatusRowSyncTimeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        __threadfence_system();
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        } else {
            wave_sync();
        }
    }
};
constexpr int plan_sync_flag_cplx(int N, int pairs) { return plan_row_spans_waves(N) ? (pairs + 3) / 4 * 4 : 0; }  // 2 ints = 1 cplx per pair, padded
template <int N>
__device__ __forceinline__ void init_row_sync(int *flags, int pairs) {
    if constexpr (plan_row_spans_waves(N)) {
        if ((int)threadIdx.x < 2 * pairs) flags[threadIdx.x] = 0;  // made visible by the block barrier of load_twiddles
    }
}

// Kernel arguments live in memory.  Left alone, the compiler loads each where it is first used and waits there, which at kernel
// entry is a chain of two or three DEPENDENT ~0.6 us round trips before the first data load goes out (phase stamps of the
// small-batch kernels: "loads issued" 1.3 us into k_pass1c_lp, 1.9 us into k_pass2c_lp).  Naming everything the kernel will need in
// one place makes it one round trip: all scalar loads are issued together, one wait.
__device__ __forceinline__ void fetch_arguments(const DeviceBuffers &b, const CascadeFrame &cf) {
    asm volatile("" ::"s"(b.h0), "s"(b.omega), "s"(b.T), "s"(b.disp), "s"(b.norm), "s"(b.foam), "s"(b.tw), "s"(b.pcol), "s"(b.rrow), "s"(b.status));
    asm volatile("" ::"s"(cf.tile_x), "s"(cf.tile_y), "s"(cf.time), "s"(cf.whitecap), "s"(cf.foam_grow_rate), "s"(cf.foam_decay), "s"(cf.cascade));
}

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
// per-wave cycle stamps of the small-batch kernels (tools/kbench_small only; the product instantiates STAMPS = false)
template <bool STAMPS>
struct WaveStamps {
    unsigned long long ts[STAMPS ? 16 : 1] = {0};
    unsigned long long rt0 = 0;
    __device__ __forceinline__ void at(int k, float keep) {
        if constexpr (STAMPS) {
            asm volatile("" ::"v"(keep));
            ts[k] = __builtin_readcyclecounter();
            if (k == 0) rt0 = __builtin_amdgcn_s_memrealtime();
        }
    }
    __device__ __forceinline__ void write(Stamp *stamps, int waves_per_block, unsigned long long tag) {
        if constexpr (STAMPS) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ts[14] = __builtin_readcyclecounter();
            ts[15] = tag;
            if ((threadIdx.x & 63) == 0) {
                Stamp st;
                for (int k = 0; k < 16; ++k) st.t[k] = ts[k];
                unsigned v;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
                st.xcc = v & 0xf;
                unsigned hw;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));  // (gfx9: cu_id [11:8], sh_id [12], se_id [15:13]: which CU of the XCD ran the wave)
                st.pad = hw;
                st.rt0 = rt0;
                st.rt1 = __builtin_amdgcn_s_memrealtime();
                stamps[(blockIdx.y * gridDim.x + blockIdx.x) * waves_per_block + threadIdx.x / 64] = st;
            }
        }
    }
};

#define OW_STAMP(k, keep)                          \
    if constexpr ((VAR & 8) != 0) {                \
        asm volatile("" ::"v"(keep));              \
        ts[k] = __builtin_readcyclecounter();      \
    }

// copy the constant twiddle table (global, plan_tw_total(N) entries) into this workgroup's LDS
template <int N>
__device__ __forceinline__ void load_twiddles(cplx *tw_lds, const cplx *__restrict__ tw) {
    for (int i = threadIdx.x; i < plan_tw_total(N); i += plan_wg_threads(N)) tw_lds[i] = tw[i];
    lds_barrier();
}

// The same in two steps: tw_fetch issues the table loads FIRST in the kernel, tw_commit (after the kernel has issued its own
// first data loads) puts them into LDS and runs the barrier.  The barrier then waits for the table only -- memory returns in
// order, and the table was asked for first -- instead of standing behind every wave's data loads.
template <int N>
struct TwPrefetch {
    static constexpr int K = (plan_tw_total(N) + plan_wg_threads(N) - 1) / plan_wg_threads(N);
    cplx v[K];
};
template <int N>
__device__ __forceinline__ void tw_fetch(TwPrefetch<N> &p, const cplx *__restrict__ tw) {
#pragma unroll
    for (int k = 0; k < TwPrefetch<N>::K; ++k) {
        const int i = (int)threadIdx.x + k * plan_wg_threads(N);
        p.v[k] = tw[i < plan_tw_total(N) ? i : 0];
    }
}
template <int N>
__device__ __forceinline__ void tw_commit(const TwPrefetch<N> &p, cplx *tw_lds) {
#pragma unroll
    for (int k = 0; k < TwPrefetch<N>::K; ++k) {
        const int i = (int)threadIdx.x + k * plan_wg_threads(N);
        if (i < plan_tw_total(N)) tw_lds[i] = p.v[k];
    }
    lds_barrier();
}

// the same for a table of COUNT entries and a block of THREADS threads (the half table of the 2048^2 pass 2 in 16- and 8-wave blocks)
template <int COUNT, int THREADS>
struct TablePrefetch {
    static constexpr int K = (COUNT + THREADS - 1) / THREADS;
    cplx v[K];
    __device__ __forceinline__ void fetch(const cplx *__restrict__ table) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int i = (int)threadIdx.x + k * THREADS;
            v[k] = table[i < COUNT ? i : 0];
        }
    }
    __device__ __forceinline__ void commit(cplx *table_lds) const {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int i = (int)threadIdx.x + k * THREADS;
            if (i < COUNT) table_lds[i] = v[k];
        }
        lds_barrier();
    }
};

// row IFFT of the 16 points in d[] (lane t of the row), exchanging through this row's LDS buffer
// (BLOCK_GATE: a workgroup barrier right before the first write into the row regions -- pass 1 uses it when
// other waves may still be draining the previous layer's staged rows out of them)
// (gate(): called right before the first write into the row regions)
// (TWH: tw is the half table of ow_device.h "HALF TABLE")
template <int N, bool TWH = false, class RS, class Gate>
__device__ __forceinline__ void row_ifft_gated(cplx *d, int t, cplx *lds_row, const cplx *__restrict__ tw, RS &rs, Gate gate) {
    fft_stage_compute<N, 0, TWH>(d, t, tw);
    gate();
    fft_stage_write<N, 0>(d, t, lds_row);
    rs.sync();
    fft_stage_read<N, 1>(d, t, lds_row);
    rs.sync();
    fft_stage_compute<N, 1, TWH>(d, t, tw);
    if constexpr (plan_S(N) == 3) {
        if constexpr (plan_lane_exchange(N)) {
            fft_lane_exchange<N>(d);  // row-swap instructions, no LDS
        } else {
            fft_stage_write<N, 1>(d, t, lds_row);
            rs.sync();
            fft_stage_read<N, 2>(d, t, lds_row);
            rs.sync();
        }
        fft_stage_compute<N, 2>(d, t, tw);
    }
}
template <int N, bool BLOCK_GATE = false, bool TWH = false, class RS>
__device__ __forceinline__ void row_ifft(cplx *d, int t, cplx *lds_row, const cplx *__restrict__ tw, RS &rs) {
    row_ifft_gated<N, TWH>(d, t, lds_row, tw, rs, [] {
        if constexpr (BLOCK_GATE) lds_barrier();
    });
}

// Pass-1 block -> (launch slot, first row).  Two things are arranged through the block index, both for speed only
// (the dispatcher places block b on XCD b % 8; nothing depends on it for correctness):
//  * the two blocks that write the two 64-byte halves of the same 128-byte lines of T (rows 16g..16g+7 and
//    16g+8..16g+15) are neighbours on one XCD, so the halves meet in one L2 before they are written back;
//  * the 16-row group g and its mirror group N/16-1-g (rows N-y) are on the same XCD next to each other: both
//    read the same h0 / omega lines (Pass1::load_modulate) and the XCD's L2 fetches them from HBM once.
// SUB > 1: the 8-row unit is shared by SUB consecutive blocks of 8 / SUB rows each (split plan with smaller blocks)
template <int N, int SUB = 1>
__device__ __forceinline__ void p1_index_to_rows(int index, int &slot, int &row0) {
    constexpr int G = N / 16;  // 16-row groups per cascade; 2*G blocks per cascade
    // (SUB > 1: the SUB blocks of one 8-row unit follow each other on the SAME XCD -- indices 8 apart --, so that their pieces of a
    //  64-byte segment of T meet in one L2)
    const int sub = (index >> 3) % SUB, b = ((index >> 3) / SUB) * 8 + (index & 7), x = b & 7, i = b >> 3;
    if constexpr (G >= 16) {
        constexpr int PER = G / 4;  // blocks per XCD per cascade: G/16 mirror pairs x 2 groups x 2 halves
        slot = i / PER;
        const int j = i % PER;
        const int pair = x + 8 * (j >> 2);
        const int g = ((j >> 1) & 1) ? G - 1 - pair : pair;
        row0 = g * 16 + (j & 1) * 8;
    } else {  // N = 128: 8 groups only
        const int grow = (x + 8 * (i >> 1)) * 16 + (i & 1) * 8;  // row index over all launch slots
        slot = grow / N;
        row0 = grow % N;
    }
    if constexpr (SUB > 1) row0 += sub * (kWgRows / SUB);
}
template <int N, int SUB = 1>
__device__ __forceinline__ void p1_block_to_rows(int &slot, int &row0) {
    p1_index_to_rows<N, SUB>((int)blockIdx.x, slot, row0);
}
template <int N>
__device__ __forceinline__ void p2_block_to_rows(int &slot, int &row0) {
    constexpr int BPC = N / kWgRows;
    slot = blockIdx.x / BPC;
    row0 = (blockIdx.x % BPC) * kWgRows;
}

template <int W>
__device__ __forceinline__ void write_stamps(const unsigned long long *ts, const DebugArgs &dbg) {
    if ((threadIdx.x & 63) == 0) {
        Stamp st;
        for (int k = 0; k < 16; ++k) st.t[k] = ts[k];
        st.xcc = xcc_id();
        st.pad = 0;
        dbg.stamps[blockIdx.x * W + threadIdx.x / 64] = st;
    }
}

// ---------------------------------------------------------------------------------------------------
// PASS 1.  One block = 8 rows.  Load + modulate, then per layer {spectrum from h, row IFFT, staged
// transposed store}.
// ---------------------------------------------------------------------------------------------------
template <int N, int VAR = 0, int AUX_T = kAuxDefault, int AUX_H = kAuxDefault>
__global__ __launch_bounds__(plan_wg_threads(N), OW_P1_WAVES) void k_pass1(DeviceBuffers buf, FrameArgs args, DebugArgs dbg) {
    constexpr int Tn = plan_T(N), P = kP;
    __shared__ __attribute__((aligned(16))) cplx lds[plan_wg_lds_cplx(N) + plan_sync_flag_cplx(N, kWgRows)];
    cplx *tw_lds = lds;  // table first: its DS offsets then fit the 16-bit immediate field
    cplx *rows_lds = lds + plan_tw_total(N);
    const int tau = threadIdx.x;
    // row inside the block (wave-uniform when a wave carries a single row) and lane inside the row
    const int rw = (Tn >= 64) ? __builtin_amdgcn_readfirstlane(tau / Tn) : tau / Tn, t = tau % Tn;
    const uint32_t plane = (uint32_t)N * N;
    cplx *lds_row = rows_lds + rw * plan_region_cplx(N);
    int *sync_flags = reinterpret_cast<int *>(lds + plan_wg_lds_cplx(N));
    RowSync<N> rs;
    rs.attach(sync_flags, rw, (tau / 64) & 1);
    rs.watch(buf.status, args.c[0].fault);
    init_row_sync<N>(sync_flags, kWgRows);
    constexpr bool kFft = (VAR & 4) == 0;
    constexpr bool kStore = (VAR & 2) == 0;
    constexpr bool kLoad = (VAR & 1) == 0;
    unsigned long long ts[(VAR & 8) ? 16 : 1] = {0};
    OW_STAMP(0, t)

    int slot, row0;
    p1_block_to_rows<N>(slot, row0);
    const CascadeFrame cf = args.c[slot];
    fetch_arguments(buf, cf);
    const int y = row0 + rw;
    const GBuf h0_c = make_gbuf(buf.h0 + (size_t)cf.cascade * plane, plane * 8u);
    const GBuf om_c = make_gbuf(buf.omega + (size_t)cf.cascade * plane, plane * 4u);
    const GBuf T_c = make_gbuf(buf.T + (size_t)slot * plane * kLayers, t_cascade_bytes(N));  // scratch: indexed by launch slot, reused by every batch

    cplx h[P];
    if constexpr (kLoad) {
        Pass1<N>::template load_modulate<AUX_H>(h, t, y, h0_c, om_c, cf.time);
    } else {
#pragma unroll
        for (int j = 0; j < P; ++j) h[j] = cplx{(float)(t + j) * 1e-3f + cf.time, (float)(t - j) * 1e-3f};
    }
    load_twiddles<N>(tw_lds, buf.tw);
    OW_STAMP(1, h[0].x)
    const float dkx = (2.0f * kPi) / cf.tile_x, dky = (2.0f * kPi) / cf.tile_y;
    const float ky = (float)(y - N / 2) * dky;
    float ik[P];
    Pass1<N>::wave_numbers(ik, t, ky, dkx);
    float keep = 0.0f;

    // Layer L's transposed rows are staged in LDS at the end of its transform and written out DURING layer L+1:
    // a quarter after each texel group of the next layer's input construction (the row regions are not touched
    // again before the next transform's first exchange, which waits behind a block barrier).  The stores of a
    // layer thus trickle out under VALU work instead of stalling all waves of the chip in one burst.
    constexpr bool kDefer = kStore;
#pragma unroll
    for (int L = 0; L < kLayers; ++L) {
        cplx d[P];
        OW_SCHED_FENCE();
        {
            // the wave-vector terms are recomputed per layer (a few VALU ops), not held across the FFTs
            const float kyo = opaque(ky), dkxo = opaque(dkx);
            const int to = opaque(t);
#pragma unroll
            for (int j = 0; j < P; ++j) opaque_inplace(h[j]);
            auto drain = [&](int g) {
                if (kDefer && L > 0) Pass1<N>::template stage_store_chunk<AUX_T>(tau, L - 1, row0, rows_lds, T_c, g);
            };
            if (L == 0) Pass1<N>::template layer_input<0>(d, h, ik, to, kyo, dkxo, drain);
            if (L == 1) Pass1<N>::template layer_input<1>(d, h, ik, to, kyo, dkxo, drain);
            if (L == 2) Pass1<N>::template layer_input<2>(d, h, ik, to, kyo, dkxo, drain);
            if (L == 3) Pass1<N>::template layer_input<3>(d, h, ik, to, kyo, dkxo, drain);
        }
        OW_SCHED_FENCE();
        OW_STAMP(2 + 3 * L, d[0].x)
        if constexpr (kFft) {
            if (kDefer && L > 0) row_ifft<N, true>(d, t, lds_row, tw_lds, rs);
            else row_ifft<N, false>(d, t, lds_row, tw_lds, rs);
        } else if (kDefer && L > 0) {
            lds_barrier();
        }
        OW_STAMP(3 + 3 * L, d[0].x)
        if (kStore || dbg.never_true) {
            rs.sync();  // the row's exchange reads are done before its region becomes the staging image
            Pass1<N>::stage_write(d, t, lds_row);
            lds_barrier();
            if (L == kLayers - 1) Pass1<N>::template stage_store<AUX_T>(tau, L, row0, rows_lds, T_c);  // nothing left to hide under
            OW_STAMP(4 + 3 * L, keep)
        } else {
#pragma unroll
            for (int j = 0; j < P; ++j) keep += d[j].x + d[j].y;
        }
    }
    if (!kStore && keep == 12345.678f) buf.T[t] = cplx{keep, keep};
    if constexpr ((VAR & 8) != 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ts[14] = __builtin_readcyclecounter();
        write_stamps<(plan_wg_threads(N) + 63) / 64>(ts, dbg);
    }
}

// ---------------------------------------------------------------------------------------------------
// PASS 2.  One block = 8 rows x' of T (the rows only share the twiddle table).  Layer order 2, 3, 1, 0
// (see Pass2 in ow_device.h).
// ---------------------------------------------------------------------------------------------------
template <int N, bool F32, int VAR = 0, int AUX_T = kAuxDefault, int AUX_O = kAuxDefault>
__global__ __launch_bounds__(plan_wg_threads(N), 4) void k_pass2(DeviceBuffers buf, FrameArgs args, DebugArgs dbg) {
    constexpr int Tn = plan_T(N), P = kP;
    __shared__ __attribute__((aligned(16))) cplx lds[plan_wg_lds_cplx(N) + plan_sync_flag_cplx(N, kWgRows)];
    cplx *tw_lds = lds;  // table first: its DS offsets then fit the 16-bit immediate field
    cplx *rows_lds = lds + plan_tw_total(N);
    const int tau = threadIdx.x;
    const int rw = (Tn >= 64) ? __builtin_amdgcn_readfirstlane(tau / Tn) : tau / Tn, t = tau % Tn;
    const uint32_t plane = (uint32_t)N * N;
    cplx *lds_row = rows_lds + rw * plan_region_cplx(N);
    int *sync_flags = reinterpret_cast<int *>(lds + plan_wg_lds_cplx(N));
    RowSync<N> rs;
    rs.attach(sync_flags, rw, (tau / 64) & 1);
    rs.watch(buf.status, args.c[0].fault);
    init_row_sync<N>(sync_flags, kWgRows);
    constexpr bool kFft = (VAR & 4) == 0;
    constexpr bool kStore = (VAR & 2) == 0;
    constexpr bool kLoad = (VAR & 1) == 0;
    unsigned long long ts[(VAR & 8) ? 16 : 1] = {0};
    OW_STAMP(0, t)

    int slot, row0;
    p2_block_to_rows<N>(slot, row0);
    const CascadeFrame cf = args.c[slot];
    fetch_arguments(buf, cf);
    const int xp = row0 + rw;
    const uint32_t tex = (uint32_t)(xp * N + t);
    const GBuf T_c = make_gbuf(buf.T + (size_t)slot * plane * kLayers, t_cascade_bytes(N));  // scratch: indexed by launch slot, reused by every batch
    const GBuf disp_c = make_gbuf(buf.disp + (size_t)cf.cascade * plane, plane * 8u);
    const GBuf norm_c = make_gbuf(buf.norm + (size_t)cf.cascade * plane, plane * 8u);
    const GBuf foam_c = make_gbuf(buf.foam + (size_t)cf.cascade * plane, plane * 2u);
    const GBuf f32_c = make_gbuf(F32 ? buf.f32 + (size_t)cf.cascade * plane * 8 : nullptr, F32 ? plane * 32u : 0u);

    float keep = 0.0f;
    auto fetch = [&](cplx *d, int layer) {
        OW_SCHED_FENCE();  // loads are issued here, not hoisted above the previous layer's transform
        if constexpr (kLoad) {
            Pass2<N>::template load_layer<AUX_T>(d, t, xp, layer, T_c);
        } else {
#pragma unroll
            for (int j = 0; j < P; ++j) d[j] = cplx{(float)(t + j + layer) * 1e-3f + cf.time, (float)(t - j) * 1e-3f};
        }
    };

    float dhx_dx[P];
    uint32_t gy_foam[P];
    {
        cplx l2[P];
        fetch(l2, 2);
        load_twiddles<N>(tw_lds, buf.tw);
        OW_STAMP(1, l2[0].x)
        if constexpr (kFft) row_ifft<N>(l2, t, lds_row, tw_lds, rs);
        cplx l3[P];
        uint32_t foam_pk[P / 2];
        fetch(l3, 3);
        if constexpr (kLoad) {
            Pass2<N>::load_foam(foam_pk, t, xp, foam_c);
        } else {
#pragma unroll
            for (int j = 0; j < P / 2; ++j) foam_pk[j] = 0;
        }
        if constexpr (kFft) row_ifft<N>(l3, t, lds_row, tw_lds, rs);
        OW_STAMP(2, l3[0].x)
        Pass2<N>::template after_layer3<F32 && kStore>(l3, l2, foam_pk, gy_foam, tex, cf, f32_c);
        if (kStore || dbg.never_true) Pass2<N>::store_foam(foam_pk, t, xp, foam_c);
#pragma unroll
        for (int o = 0; o < P; ++o) dhx_dx[o] = l2[OutMap<N>::slot_of(o)].y;
    }
    float hz[P];
    {
        cplx l1[P];
        fetch(l1, 1);
        if constexpr (kFft) row_ifft<N>(l1, t, lds_row, tw_lds, rs);
        if (kStore || dbg.never_true) {
            Pass2<N>::template after_layer1<F32, AUX_O>(l1, dhx_dx, gy_foam, tex, norm_c, f32_c);
        } else {
#pragma unroll
            for (int j = 0; j < P; ++j) keep += l1[j].y + dhx_dx[j] + (float)gy_foam[j];
        }
#pragma unroll
        for (int o = 0; o < P; ++o) hz[o] = l1[OutMap<N>::slot_of(o)].x;
    }
    {
        cplx l0[P];
        fetch(l0, 0);
        if constexpr (kFft) row_ifft<N>(l0, t, lds_row, tw_lds, rs);
        OW_STAMP(3, l0[0].x)
        if (kStore || dbg.never_true) {
            Pass2<N>::template after_layer0<F32, AUX_O>(l0, hz, t, xp, tex, disp_c, f32_c);
        } else {
#pragma unroll
            for (int j = 0; j < P; ++j) keep += l0[j].x + l0[j].y + hz[j];
        }
    }
    if (!kStore && keep == 12345.678f) buf.disp[t] = u16x4{1, 2, 3, 4};
    if constexpr ((VAR & 8) != 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ts[14] = __builtin_readcyclecounter();
        write_stamps<(plan_wg_threads(N) + 63) / 64>(ts, dbg);
    }
}


// ===================================================================================================
// COMPACT-INTERMEDIATE variants of the two standard kernels (N >= 256).  Three packed
// layers cross T instead of four (24 instead of 32 B/texel each way): see Pass1::layer_input_c / Pass2::derive_dx in
// ow_device.h and tests/test_compact_math.py for the algebra.  Same structure as k_pass1 / k_pass2 otherwise.
// ===================================================================================================
// STAMPS (tools/kbench only): per-wave cycle stamps at the phase boundaries, written to `stamps`
// The compact pass 1 of ONE item = 8 consecutive rows row0 .. row0 + 7 of launch slot `tslot` (scratch) / cascade cf.cascade,
// all layers, as a function of the lane index tau (0 .. 8 * N/16 - 1) inside the item: shared by k_pass1c (one item per block) and
// by the tick-group kernel (k_tick_group_c_lp: items of several ticks, several side by side in one block).  issued() is called
// right after the item's global loads have gone out (the stand-alone kernel commits its twiddle prefetch there); the block barriers
// inside are block-wide: every item of a block must come from the same half of the rows (same layer sequence).
template <int N, int AUX_T, int AUX_H, class Issued, class Stamper>
__device__ __forceinline__ void pass1c_item(const DeviceBuffers &buf, const CascadeFrame &cf, float time, int tslot, int row0, int tau, cplx *tw_lds,
                                            cplx *rows_lds, RowSync<N> &rs, Issued issued, Stamper stamp) {
    constexpr int Tn = plan_T(N), P = kP, LC = Pass1<N>::kCompactLayers;
    const int rw = (Tn >= 64) ? __builtin_amdgcn_readfirstlane(tau / Tn) : tau / Tn, t = tau % Tn;
    const uint32_t plane = (uint32_t)N * N;
    cplx *lds_row = rows_lds + rw * plan_region_cplx(N);
    const int y = row0 + rw;
    const GBuf h0_c = make_gbuf(buf.h0 + (size_t)cf.cascade * plane, plane * 8u);
    const GBuf om_c = make_gbuf(buf.omega + (size_t)cf.cascade * plane, plane * 4u);
    const GBuf T_c = make_gbuf(buf.T + (size_t)tslot * plane * kLayers, t_cascade_bytes(N));
    const GBuf pcol_c = make_gbuf(buf.pcol + (size_t)tslot * N, (uint32_t)N * 8u);
    const GBuf rrow_c = make_gbuf(buf.rrow + (size_t)tslot * N * 4, (uint32_t)N * 32u);

    cplx h[P];
    const float dkx = (2.0f * kPi) / cf.tile_x, dky = (2.0f * kPi) / cf.tile_y;
    const float ky = (float)(y - N / 2) * dky;
    float ik[P];
    {
        cplx a[P], b[P];
        float om[P];
        Pass1<N>::template load_raw<AUX_H>(a, b, om, t, y, h0_c, om_c);
        issued();
        Pass1<N>::modulate(h, a, b, om, time);
    }
    stamp(1, h[0].x);
    Pass1<N>::wave_numbers(ik, t, ky, dkx);
    if (t == 0) gstore8(pcol_c, Pass2<N>::pcol_index(y) * 8u, 0u, Pass1<N>::column_term(h, ik, t, dkx));

    // the wave that holds texel row 0 (its first lane does; for N < 1024 it holds a few more rows, transformed along and
    // discarded) does the three extra transforms of that row, straight to the side buffer
    if (__builtin_amdgcn_readfirstlane(y) == 0) {
#pragma unroll
        for (int Q = 1; Q <= 3; ++Q) {
            cplx d[P];
            OW_SCHED_FENCE();
            {
                const float kyo = opaque(ky), dkxo = opaque(dkx);
                const int to = opaque(t);
#pragma unroll
                for (int j = 0; j < P; ++j) opaque_inplace(h[j]);
                if (Q == 1) Pass1<N>::template row0_input<1>(d, h, ik, to, kyo, dkxo);
                if (Q == 2) Pass1<N>::template row0_input<2>(d, h, ik, to, kyo, dkxo);
                if (Q == 3) Pass1<N>::template row0_input<3>(d, h, ik, to, kyo, dkxo);
            }
            OW_SCHED_FENCE();
            row_ifft<N, false>(d, t, lds_row, tw_lds, rs);
            rs.sync();  // the row region is free again before the next transform writes into it
            if (y == 0) {
#pragma unroll
                for (int o = 0; o < P; ++o) gstore8(rrow_c, (uint32_t)(t + Tn * o) * 32u, (uint32_t)Q * 8u, d[OutMap<N>::slot_of(o)]);
            }
        }
    }

    // rows below N/2 skip layer 1: hz of row N - y is the conjugate of row y's (block-uniform: a block's rows lie in one half)
    const bool lower = row0 < N / 2;
#pragma unroll
    for (int L = 0; L < LC; ++L) {
        if (L == 1 && lower) continue;
        cplx d[P];
        OW_SCHED_FENCE();
        {
            const float kyo = opaque(ky), dkxo = opaque(dkx);
            const int to = opaque(t);
#pragma unroll
            for (int j = 0; j < P; ++j) opaque_inplace(h[j]);
            auto drain = [&](int g) {  // the previous layer that was staged: L - 1, or 0 when layer 1 was skipped
                if (L > 0) Pass1<N>::template stage_store_chunk<AUX_T>(tau, (L == 2 && lower) ? 0 : L - 1, row0, rows_lds, T_c, g);
            };
            if (L == 0) Pass1<N>::template layer_input_c<0>(d, h, ik, to, kyo, dkxo, drain);
            if (L == 1) Pass1<N>::template layer_input_c<1>(d, h, ik, to, kyo, dkxo, drain);
            if (L == 2) Pass1<N>::template layer_input_c<2>(d, h, ik, to, kyo, dkxo, drain);
        }
        OW_SCHED_FENCE();
        stamp(2 + 3 * L, d[0].x);
        if (L > 0) row_ifft<N, true>(d, t, lds_row, tw_lds, rs);
        else row_ifft<N, false>(d, t, lds_row, tw_lds, rs);
        stamp(3 + 3 * L, d[0].x);
        rs.sync();
        Pass1<N>::stage_write(d, t, lds_row);
        lds_barrier();
        if (L == LC - 1) Pass1<N>::template stage_store<AUX_T>(tau, L, row0, rows_lds, T_c);
        stamp(4 + 3 * L, d[0].x);
    }
}

template <int N, int AUX_T = kAuxDefault, int AUX_H = kAuxDefault, bool STAMPS = false>
__global__ __launch_bounds__(plan_wg_threads(N), OW_P1_WAVES) void k_pass1c(DeviceBuffers buf, FrameArgs args, Stamp *stamps = nullptr) {
    unsigned long long ts[STAMPS ? 16 : 1] = {0};
    auto stamp = [&](int k, float keep) {
        if constexpr (STAMPS) {
            asm volatile("" ::"v"(keep));
            ts[k] = __builtin_readcyclecounter();
        }
    };
    stamp(0, 0.0f);
    constexpr int Tn = plan_T(N);
    static_assert(Tn >= 16, "Pass2::load_c1 needs N/16 to be a multiple of the 16-row line");
    __shared__ __attribute__((aligned(16))) cplx lds[plan_wg_lds_cplx(N) + plan_sync_flag_cplx(N, kWgRows)];
    cplx *tw_lds = lds;
    cplx *rows_lds = lds + plan_tw_total(N);
    const int tau = threadIdx.x;
    int *sync_flags = reinterpret_cast<int *>(lds + plan_wg_lds_cplx(N));
    RowSync<N> rs;
    rs.attach(sync_flags, tau / Tn, (tau / 64) & 1);
    rs.watch(buf.status, args.c[0].fault);
    init_row_sync<N>(sync_flags, kWgRows);

    int slot, row0;
    p1_block_to_rows<N>(slot, row0);
    const CascadeFrame cf = args.c[slot];
    fetch_arguments(buf, cf);
    TwPrefetch<N> twp;
    tw_fetch<N>(twp, buf.tw);
    pass1c_item<N, AUX_T, AUX_H>(buf, cf, cf.time, slot, row0, tau, tw_lds, rows_lds, rs, [&] { tw_commit<N>(twp, tw_lds); }, stamp);
    if constexpr (STAMPS) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ts[14] = __builtin_readcyclecounter();
        ts[15] = (unsigned long long)row0;
        if ((threadIdx.x & 63) == 0) {
            Stamp st;
            for (int k = 0; k < 16; ++k) st.t[k] = ts[k];
            st.xcc = xcc_id();
            st.pad = 0;
            stamps[blockIdx.x * ((plan_wg_threads(N) + 63) / 64) + threadIdx.x / 64] = st;
        }
    }
}

// The compact pass 2 of ONE item = 8 consecutive columns row0 .. row0 + 7 of launch slot `tslot` (scratch) / cascade cf.cascade, as
// a function of the lane index tau inside the item: shared by k_pass2c and the compact tick-group kernel (k_tick_group_c, where a
// block walks through several ticks of its columns).  foam_pk: the lane's 16 FP16 foam values; foam_io & 1: load them from the
// foam plane first, & 2: store them back at the end (the group kernel carries them in registers from tick to tick).
// At N = 2048 (rows of two waves) tw_lds is the HALF table (p2c_table_total(N) entries from p2c_table(buf)): the row's second wave rotates
// its stage-0 twiddles by a constant instead of reading its own half of a 16 KB table -- so that a 4-column block fits a CU twice.
constexpr bool p2c_half_table(int N) { return plan_row_spans_waves(N); }
constexpr int p2c_table_total(int N) { return p2c_half_table(N) ? plan_twh_total(N) : plan_tw_total(N); }
template <int N>
__device__ __forceinline__ const cplx *p2c_table(const DeviceBuffers &buf) { return p2c_half_table(N) ? buf.tw_half : buf.tw; }
template <int N, bool F32, int AUX_T, int AUX_O, class RS, class Issued>
__device__ __forceinline__ void pass2c_item(const DeviceBuffers &buf, const CascadeFrame &cf, int tslot, int row0, int tau, cplx *tw_lds, cplx *rows_lds,
                                            RS &rs, Issued issued, uint32_t (&foam_pk)[kP / 2], int foam_io = 3) {
    constexpr int Tn = plan_T(N), P = kP;
    constexpr bool TWH = p2c_half_table(N);
    const int rw = (Tn >= 64) ? __builtin_amdgcn_readfirstlane(tau / Tn) : tau / Tn, t = tau % Tn;
    const uint32_t plane = (uint32_t)N * N;
    cplx *lds_row = rows_lds + rw * plan_region_cplx(N);
    const int xp = row0 + rw;
    const GBuf T_c = make_gbuf(buf.T + (size_t)tslot * plane * kLayers, t_cascade_bytes(N));
    const GBuf pcol_c = make_gbuf(buf.pcol + (size_t)tslot * N, (uint32_t)N * 8u);
    const GBuf rrow_c = make_gbuf(buf.rrow + (size_t)tslot * N * 4, (uint32_t)N * 32u);
    const GBuf disp_c = make_gbuf(buf.disp + (size_t)cf.cascade * plane, plane * 8u);
    const GBuf norm_c = make_gbuf(buf.norm + (size_t)cf.cascade * plane, plane * 8u);
    const GBuf foam_c = make_gbuf(buf.foam + (size_t)cf.cascade * plane, plane * 2u);
    const GBuf f32_c = make_gbuf(F32 ? buf.f32 + (size_t)cf.cascade * plane * 8 : nullptr, F32 ? plane * 32u : 0u);
    const float dky = (2.0f * kPi) / cf.tile_y;
    // texel row 0's three transforms at this x' (ky-index 0 of F1..F3): entry q of the side buffer
    auto side_row = [&](int q) { return gload8(rrow_c, (uint32_t)xp * 32u, (uint32_t)q * 8u); };

    uint32_t hz_pk[P / 2];
    float c2[P];
    {
        cplx f2[P];
        OW_SCHED_FENCE();
        Pass2<N>::template load_c1<AUX_T>(f2, t, xp, dky, T_c);
        const cplx r2 = side_row(2);
        issued();
        Pass2<N>::put_row0(f2, t, r2);
        row_ifft<N, false, TWH>(f2, opaque(t), lds_row, tw_lds, rs);
        OW_SCHED_FENCE();
        Pass2<N>::template after_f2<F32>(f2, hz_pk, c2, (uint32_t)(xp * N + opaque(t)), f32_c);
    }
    cplx c0[P];  // C0, loaded once: transformed as it is (F0) and, from this copy, as i ky C0 + P (F1)
    {
        cplx f0[P];
        OW_SCHED_FENCE();
        const int tq = opaque(t);  // per phase: offsets derived from the lane index are recomputed, not carried through the transforms
        Pass2<N>::template load_layer<AUX_T>(c0, tq, xp, 0, T_c);
#pragma unroll
        for (int j = 0; j < P; ++j) f0[j] = c0[j];
        row_ifft<N, false, TWH>(f0, opaque(t), lds_row, tw_lds, rs);
        OW_SCHED_FENCE();
        const int tr = opaque(t);
        Pass2<N>::template after_f0<F32, AUX_O>(f0, hz_pk, tr, xp, (uint32_t)(xp * N + tr), disp_c, f32_c);
    }
    float dhx_dx[P];
    uint32_t gx_pk[P / 2];
    {
        OW_SCHED_FENCE();
        const int tq = opaque(t);
        Pass2<N>::derive_dx(c0, tq, xp, dky, pcol_c);
        Pass2<N>::put_row0(c0, tq, side_row(1));
        row_ifft<N, false, TWH>(c0, opaque(t), lds_row, tw_lds, rs);
        OW_SCHED_FENCE();
        Pass2<N>::template after_f1<F32>(c0, dhx_dx, gx_pk, (uint32_t)(xp * N + opaque(t)), f32_c);
    }
    {
        cplx f3[P];
        OW_SCHED_FENCE();
        const int tq = opaque(t);
        Pass2<N>::template load_layer<AUX_T>(f3, tq, xp, 2, T_c);
        Pass2<N>::put_row0(f3, tq, side_row(3));
        if (foam_io & 1) Pass2<N>::load_foam(foam_pk, tq, xp, foam_c);
        row_ifft<N, false, TWH>(f3, opaque(t), lds_row, tw_lds, rs);
        OW_SCHED_FENCE();
        const int tr = opaque(t);
        Pass2<N>::template after_f3<F32, AUX_O>(f3, dhx_dx, c2, gx_pk, foam_pk, (uint32_t)(xp * N + tr), cf, norm_c, f32_c);
        if (foam_io & 2) Pass2<N>::store_foam(foam_pk, tr, xp, foam_c);
    }
}

template <int N, bool F32, int AUX_T = kAuxDefault, int AUX_O = kAuxDefault>
__global__ __launch_bounds__(plan_wg_threads(N), 4) void k_pass2c(DeviceBuffers buf, FrameArgs args) {
    constexpr int Tn = plan_T(N);
    static_assert(Tn >= 16, "Pass2::load_c1 needs N/16 to be a multiple of the 16-row line");
    constexpr int TW = p2c_table_total(N), ROWS_CPLX = plan_region_cplx(N) * kWgRows;
    __shared__ __attribute__((aligned(16))) cplx lds[TW + ROWS_CPLX + plan_sync_flag_cplx(N, kWgRows)];
    cplx *tw_lds = lds;
    cplx *rows_lds = lds + TW;
    const int tau = threadIdx.x;
    int *sync_flags = reinterpret_cast<int *>(lds + TW + ROWS_CPLX);
    RowSync<N> rs;
    rs.attach(sync_flags, tau / Tn, (tau / 64) & 1);
    rs.watch(buf.status, args.c[0].fault);
    init_row_sync<N>(sync_flags, kWgRows);

    int slot, row0;
    p2_block_to_rows<N>(slot, row0);
    const CascadeFrame cf = args.c[slot];
    fetch_arguments(buf, cf);
    TablePrefetch<TW, plan_wg_threads(N)> twp;
    twp.fetch(p2c_table<N>(buf));
    uint32_t foam_pk[kP / 2];
    pass2c_item<N, F32, AUX_T, AUX_O>(buf, cf, slot, row0, tau, tw_lds, rows_lds, rs, [&] { twp.commit(tw_lds); }, foam_pk);
}

// ===================================================================================================
// SPLIT-PLAN compact pass 1 for N = 2048 (see "SPLIT PLAN" in ow_device.h): a row is two waves, one per parity of the element
// index, each running the N/2 plan on its own; the radix-2 step that joins them is done by the storing threads, so the layer
// transforms need no rendezvous between the waves of a row.  Block = ROWS rows = 2 ROWS waves; wave 2r + w = parity w of row r.
// Twiddles: buf.tw_split = [table of the N/2 plan][W_N^k, k = 0 .. N/2 - 1].
// (Measured on the same box: 33.5 us per 2048^2 cascade against 36.0 for k_pass1c<2048>, whose rows exchange across their two
// waves four times per transform; x 4: 39.4 against 42.2.)
// ROWS = 4 (OW_SPLIT_P1_ROWS): two independent 8-wave blocks per CU (74 KB of LDS each) instead of one 16-wave block whose block
// barriers march the whole CU through load -> transform -> store in step.  A block then writes 32-byte halves of T's 64-byte
// segments, and that only pays when the two blocks of an 8-row unit run on the SAME XCD, one after the other, so that the halves
// meet in one L2 (p1_index_to_rows: indices 8 apart): 34.1 -> 29.3 - 30.9 us per 2048^2 cascade, 2048^2 x 4 277 -> 252 - 254 us per
// tick.  (With the halves on neighbouring XCDs -- consecutive block indices, round 2's first attempt -- 4-row blocks lost; 2-row
// blocks, 16-byte pieces: 39.7 us.  profiles/r02_2048_split_plan.txt)
// ===================================================================================================
#ifndef OW_SPLIT_P1_ROWS
#define OW_SPLIT_P1_ROWS 4
#endif
template <int N, int ROWS = kWgRows>
struct SplitGeo {
    static constexpr int H = N / 2, TH = plan_T(H), R = plan_region_cplx(H), TW = plan_tw_total(H);
    static constexpr int kRows = ROWS;                       // rows per block
    static constexpr int kThreads = 2 * ROWS * TH;           // 1024 for 8 rows
    static constexpr int kLdsCplx = TW + 2 * ROWS * R;       // table + virtual-row regions (E regions of the rows, then O regions)
    static constexpr int kTwPerThread = (TW + kThreads - 1) / kThreads;
};
// the N/2-plan table into LDS: one entry per thread, asked for before the kernel's first data loads, committed after them
template <class SG>
struct SplitTw {
    cplx v[SG::kTwPerThread];
};
template <class SG>
__device__ __forceinline__ void split_tw_fetch(SplitTw<SG> &p, const cplx *__restrict__ tw_split, int tau) {
#pragma unroll
    for (int k = 0; k < SG::kTwPerThread; ++k) {
        const int i = tau + k * SG::kThreads;
        p.v[k] = tw_split[i < SG::TW ? i : 0];
    }
}
template <class SG>
__device__ __forceinline__ void split_tw_commit(const SplitTw<SG> &p, cplx *tw_lds, int tau) {
#pragma unroll
    for (int k = 0; k < SG::kTwPerThread; ++k) {
        const int i = tau + k * SG::kThreads;
        if (i < SG::TW) tw_lds[i] = p.v[k];
    }
    lds_barrier();
}

// one ITEM of the split-plan pass 1 = rows row0 .. row0 + ROWS - 1 of launch slot `tslot` (scratch) / cascade cf.cascade at time `time`, as a
// function of the lane index tau (0 .. SG::kThreads - 1); the item is the whole block (its barriers are the block's).  tw_lds: the N/2-plan
// table (SG::TW complex); rows_lds: 2 ROWS row regions (SG::R complex each); sync_flags: 2 ROWS ints.  stamp(k, keep): developer phase stamps
// (tools/kbench_2048pair).
template <int N, int ROWS, int AUX_T, int AUX_H, class Stamper>
__device__ __forceinline__ void pass1c_split_item(const DeviceBuffers &buf, const CascadeFrame &cf, float time, int tslot, int row0, int fault, cplx *tw_lds,
                                                  cplx *rows_lds, int *sync_flags, int tau, Stamper stamp) {
    using SG = SplitGeo<N, ROWS>;
    constexpr int H = SG::H, TH = SG::TH, P = kP, LC = Pass1<N>::kCompactLayers, T = plan_T(N);
    const int wv = __builtin_amdgcn_readfirstlane(tau / 64), rw = wv >> 1, w = wv & 1;  // row inside the item, parity
    const int tp = tau % 64, t = 2 * tp + w;                                              // physical lane, logical lane of the N plan
    cplx *vrow = rows_lds + (w * ROWS + rw) * SG::R;
    const uint32_t plane = (uint32_t)N * N;
    RowSync<H> rs;    // inside one wave: compiler ordering only
    RowSync<N> pair;  // the two waves of a row (used by the pair that owns texel row 0 only)
    pair.attach(sync_flags, rw, w);
    pair.watch(buf.status, fault);
    if (tau < 2 * ROWS) sync_flags[tau] = 0;  // made visible by the block barrier of the twiddle commit

    const int y = row0 + rw;
    const GBuf h0_c = make_gbuf(buf.h0 + (size_t)cf.cascade * plane, plane * 8u);
    const GBuf om_c = make_gbuf(buf.omega + (size_t)cf.cascade * plane, plane * 4u);
    const GBuf T_c = make_gbuf(buf.T + (size_t)tslot * plane * kLayers, t_cascade_bytes(N));
    const GBuf pcol_c = make_gbuf(buf.pcol + (size_t)tslot * N, (uint32_t)N * 8u);
    const GBuf rrow_c = make_gbuf(buf.rrow + (size_t)tslot * N * 4, (uint32_t)N * 32u);

    // storing thread: row q of the block, x' = xi + T m (and + N/2): W_N^xi from the table, the ordinal part is compile time
    const int q = tau % ROWS, xi = tau / ROWS;  // xi in [0, T)
    cplx h[P];
    cplx wxi;
    const float dkx = (2.0f * kPi) / cf.tile_x, dky = (2.0f * kPi) / cf.tile_y;
    const float ky = (float)(y - N / 2) * dky;
    float ik[P];
    {
        SplitTw<SG> twv;
        split_tw_fetch<SG>(twv, buf.tw_split, tau);
        wxi = buf.tw_split[SG::TW + xi];
        cplx a[P], b[P];
        float om[P];
        Pass1<N>::template load_raw<AUX_H>(a, b, om, t, y, h0_c, om_c);
        stamp(1, 0.0f);  // loads issued
        split_tw_commit<SG>(twv, tw_lds, tau);
        stamp(2, a[15].x + om[15]);  // table committed (block barrier), own data arrived
        Pass1<N>::modulate(h, a, b, om, time);
    }
    stamp(3, h[15].x);  // modulated
    Pass1<N>::wave_numbers(ik, t, ky, dkx);
    if (t == 0) gstore8(pcol_c, Pass2<N>::pcol_index(y) * 8u, 0u, Pass1<N>::column_term(h, ik, t, dkx));

    // E[k] +- W_N^k O[k] for k = xi + T m, m = 2g and 2g + 1 (chunk g of four): staged values of row q
    const cplx *e_reg = rows_lds + q * SG::R, *o_reg = rows_lds + (ROWS + q) * SG::R;
    auto combine = [&](int m, cplx &lo, cplx &hi) {
        const cplx e = lds_read(e_reg + xi + T * m), o = lds_read(o_reg + xi + T * m);
        // W_N^(xi + T m) = W_N^xi * exp(2 pi i m / 16): two packed instructions where it is used.  (The base is made opaque so that
        // the eight products are NOT kept in registers from one layer to the next: 14 VGPRs this kernel does not have.)
        cplx wx = wxi;
        opaque_inplace(wx);
        cplx tw = wx;
        switch (m) {
            case 1: tw = cmul_const(wx, root32_cos(2), root32_sin(2)); break;
            case 2: tw = cmul_const(wx, root32_cos(4), root32_sin(4)); break;
            case 3: tw = cmul_const(wx, root32_cos(6), root32_sin(6)); break;
            case 4: tw = cmuli(wx); break;
            case 5: tw = cmul_const(wx, root32_cos(10), root32_sin(10)); break;
            case 6: tw = cmul_const(wx, root32_cos(12), root32_sin(12)); break;
            case 7: tw = cmul_const(wx, root32_cos(14), root32_sin(14)); break;
            default: break;
        }
        const cplx z = cmul(o, tw);
        lo = cadd(e, z);
        hi = csub(e, z);
    };
    const uint32_t voff = t_unit(N, 0, xi, row0 + q) * 8u;
    auto store_chunk = [&](int layer, int g) {  // x' = xi + T m and xi + T (m + 8), m = 2g, 2g + 1
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
            const int m = 2 * g + mm;
            cplx lo, hi;
            combine(m, lo, hi);
            gstore8<AUX_T>(T_c, voff, (t_unit(N, layer, 0, 0) + t_unit(N, 0, T * m, 0)) * 8u, lo);
            gstore8<AUX_T>(T_c, voff, (t_unit(N, layer, 0, 0) + t_unit(N, 0, T * (m + 8), 0)) * 8u, hi);
            OW_SCHED_FENCE();
        }
    };

    // The two waves that hold texel row 0 do the three extra transforms of that row between themselves, straight to the side
    // buffer: they stage E / O in their own regions, meet (the only rendezvous of this kernel: one pair per cascade), and each
    // lane joins and stores the sixteen x' = (tp + TH w) + T m, m = 0..15.  No block barrier is involved, so the other fourteen
    // waves of the block are not held up (measured: done by the whole block around block barriers, this tail cost 8 us per launch).
    if (__builtin_amdgcn_readfirstlane(y) == 0) {
        const int xr = tp + TH * w;  // this lane's x' (mod T) in the join
        const cplx wxr = buf.tw_split[SG::TW + xr];
        const cplx *e0 = rows_lds + rw * SG::R, *o0 = rows_lds + (ROWS + rw) * SG::R;
#pragma unroll
        for (int Q = 1; Q <= 3; ++Q) {
            cplx d[P];
            OW_SCHED_FENCE();
            {
                const float kyo = opaque(ky), dkxo = opaque(dkx);
                const int to = opaque(t);
#pragma unroll
                for (int j = 0; j < P; ++j) opaque_inplace(h[j]);
                if (Q == 1) Pass1<N>::template row0_input<1>(d, h, ik, to, kyo, dkxo);
                if (Q == 2) Pass1<N>::template row0_input<2>(d, h, ik, to, kyo, dkxo);
                if (Q == 3) Pass1<N>::template row0_input<3>(d, h, ik, to, kyo, dkxo);
            }
            OW_SCHED_FENCE();
            row_ifft<H, false>(d, tp, vrow, tw_lds, rs);
            rs.sync();
#pragma unroll
            for (int o = 0; o < P; ++o) vrow[tp + TH * o] = d[OutMap<H>::slot_of(o)];
            pair.sync();
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const cplx e = lds_read(e0 + xr + T * m), o = lds_read(o0 + xr + T * m);
                cplx wx = wxr;  // (opaque: the eight products are formed where they are used, not kept across the three transforms)
                opaque_inplace(wx);
                const cplx z = cmul(o, cmul_const(wx, root32_cos(2 * m), root32_sin(2 * m)));
                gstore8(rrow_c, (uint32_t)(xr + T * m) * 32u, (uint32_t)Q * 8u, cadd(e, z));
                gstore8(rrow_c, (uint32_t)(xr + T * (m + 8)) * 32u, (uint32_t)Q * 8u, csub(e, z));
                OW_SCHED_FENCE();
            }
            pair.sync();  // both waves have read the regions: free for the next transform
        }
    }

    const bool lower = row0 < N / 2;  // rows below N/2 skip layer 1 (block-uniform)
#pragma unroll
    for (int L = 0; L < LC; ++L) {
        if (L == 1 && lower) continue;
        cplx d[P];
        OW_SCHED_FENCE();
        {
            const float kyo = opaque(ky), dkxo = opaque(dkx);
            const int to = opaque(t);
#pragma unroll
            for (int j = 0; j < P; ++j) opaque_inplace(h[j]);
            auto drain = [&](int g) {  // the previous layer that was staged: L - 1, or 0 when layer 1 was skipped
                if (L > 0) store_chunk((L == 2 && lower) ? 0 : L - 1, g);
            };
            if (L == 0) Pass1<N>::template layer_input_c<0>(d, h, ik, to, kyo, dkxo, drain);
            if (L == 1) Pass1<N>::template layer_input_c<1>(d, h, ik, to, kyo, dkxo, drain);
            if (L == 2) Pass1<N>::template layer_input_c<2>(d, h, ik, to, kyo, dkxo, drain);
        }
        OW_SCHED_FENCE();
        stamp(4 + 3 * L, d[0].x);  // layer input built (previous layer's stores issued)
        if (L > 0) row_ifft<H, true>(d, tp, vrow, tw_lds, rs);  // (block gate: the staged rows have been drained by every wave)
        else row_ifft<H, false>(d, tp, vrow, tw_lds, rs);
        stamp(5 + 3 * L, d[0].x);  // transformed
        rs.sync();
#pragma unroll
        for (int o = 0; o < P; ++o) vrow[tp + TH * o] = d[OutMap<H>::slot_of(o)];
        lds_barrier();
        stamp(6 + 3 * L, 0.0f);  // staged, block barrier passed
        if (L == LC - 1) {
#pragma unroll
            for (int g = 0; g < 4; ++g) store_chunk(L, g);
        }
    }
    stamp(13, 0.0f);  // last stores issued
}

template <int N, int ROWS = OW_SPLIT_P1_ROWS, int AUX_T = kAuxDefault, int AUX_H = kAuxDefault, bool STAMPS = false>
__global__ __launch_bounds__((SplitGeo<N, ROWS>::kThreads), 4) void k_pass1c_split(DeviceBuffers buf, FrameArgs args, Stamp *stamps = nullptr) {
    using SG = SplitGeo<N, ROWS>;
    WaveStamps<STAMPS> ws;
    ws.at(0, 0.0f);
    __shared__ __attribute__((aligned(16))) cplx lds[SG::kLdsCplx + plan_sync_flag_cplx(N, ROWS)];
    int slot, row0;
    p1_block_to_rows<N, kWgRows / ROWS>(slot, row0);
    const CascadeFrame cf = args.c[slot];
    fetch_arguments(buf, cf);
    pass1c_split_item<N, ROWS, AUX_T, AUX_H>(buf, cf, cf.time, slot, row0, args.c[0].fault, lds, lds + SG::TW, reinterpret_cast<int *>(lds + SG::kLdsCplx),
                                             (int)threadIdx.x, [&](int k, float keep) { ws.at(k, keep); });
    ws.write(stamps, SG::kThreads / 64, (unsigned long long)row0);
}

// ===================================================================================================
// TICK PAIRS at N = 2048 (k_tick_pair_c_split): one launch = pass 2 of one cascade and pass 1 of the NEXT cascade of the stream (the tick's
// next cascade, or the first cascade one tick later), both in 8-WAVE blocks: pass 1 in the split plan's 4-row items (k_pass1c_split's own
// block), pass 2 in blocks of 4 columns (k_pass2c's item body on half as many rows).  Either kind takes 78 KB of LDS -- for pass 2 that needs
// the HALF twiddle table (ow_device.h): with the full 16 KB table a 4-column block is 86 KB and a CU holds one -- so a CU holds TWO blocks, of
// either kind, and the chunks of 8 blocks (one per XCD) alternate between the passes: every CU runs a pass-1 block beside a pass-2 block,
// one in its transforms while the other waits for memory.  That is what the 1024^2 pair kernel lives on, and what the first form of this
// kernel (16-wave blocks, one per CU: profiles/r04_2048_pair_v1_16wave_blocks_rejected.txt) lacked.  One cascade's compact intermediate
// (84 MB) is written by one launch and read by the next while still resident in the Infinity Cache.  Same item bodies as k_pass1c_split /
// k_pass2c: bit-identical results (tests/test_tick_groups.py).
// ===================================================================================================
template <int N>
struct PairSplitGeo {
    using SG = SplitGeo<N, 4>;
    static constexpr int kThreads = SG::kThreads;                     // 512: a 4-row split item = 8 waves
    static constexpr int kCols = kThreads / plan_T(N);                // columns of a pass-2 block: 4
    static constexpr int kP1Cplx = SG::kLdsCplx + plan_sync_flag_cplx(N, 4);
    static constexpr int kP2Tw = p2c_table_total(N);
    static constexpr int kP2Cplx = kP2Tw + kCols * plan_region_cplx(N) + plan_sync_flag_cplx(N, kCols);
    static constexpr int kLdsCplx = kP1Cplx > kP2Cplx ? kP1Cplx : kP2Cplx;
    static_assert(kCols * plan_T(N) == kThreads && 2 * kLdsCplx * (int)sizeof(cplx) <= 160 * 1024, "two blocks per CU, of either kind");
};
// (the two tick-pair kernels take PairArgs: 360 bytes of kernarg with the buffers, where the tick groups' FrameArgs + TickGroupArgs are 1 052;
//  same box, alternating builds: 52.08 -> 51.78 us per tick at 1024^2 x 4, profiles/r05_ab_rounds.txt)
template <int N, bool F32, bool STAMPS = false>
__global__ __launch_bounds__((PairSplitGeo<N>::kThreads), 4) void k_tick_pair_c_split(DeviceBuffers buf, PairArgs g, Stamp *stamps = nullptr) {
    static_assert(plan_split(N), "rows that span two waves (N = 2048)");
    using PG = PairSplitGeo<N>;
    using SG = typename PG::SG;
    WaveStamps<STAMPS> ws;
    ws.at(0, 0.0f);
    __shared__ __attribute__((aligned(16))) cplx lds[PG::kLdsCplx];
    // g.n2 pass-2 blocks and g.n1 pass-1 blocks (multiples of 8, either may be 0): alternate in chunks of 8 while both last
    int index = blockIdx.x;
    bool first;  // is this a pass-1 block?
    {
        const int both = 2 * (g.n2 < g.n1 ? g.n2 : g.n1), chunk = index >> 3;
        if (index < both) {
            first = chunk & 1;
            index = ((chunk >> 1) << 3) + (index & 7);
        } else {
            first = g.n2 < g.n1;
            index -= both / 2;
        }
    }
    if (!first) {  // ---- pass 2 of 4 columns ----
        cplx *tw_lds = lds;
        cplx *rows_lds = lds + PG::kP2Tw;
        const int tau = threadIdx.x;
        int *sync_flags = reinterpret_cast<int *>(lds + PG::kP2Tw + PG::kCols * plan_region_cplx(N));
        RowSync<N> rs;
        rs.attach(sync_flags, tau / plan_T(N), (tau / 64) & 1);
        rs.watch(buf.status, g.fault);
        init_row_sync<N>(sync_flags, PG::kCols);
        constexpr int BPC = N / PG::kCols;
        const int slot = index / BPC, row0 = (index % BPC) * PG::kCols;
        const CascadeFrame cf = pair_frame(g, g.first2 + slot);
        fetch_arguments(buf, cf);
        TablePrefetch<PG::kP2Tw, PG::kThreads> twp;
        twp.fetch(p2c_table<N>(buf));
        uint32_t foam_pk[kP / 2];
        pass2c_item<N, F32, kAuxDefault, kAuxNT>(buf, cf, g.tbase2 + slot, row0, tau, tw_lds, rows_lds, rs, [&] { twp.commit(tw_lds); }, foam_pk);
        ws.write(stamps, PG::kThreads / 64, 100000ull);
        return;
    }
    // ---- pass 1 of 4 rows: k_pass1c_split's block (the two blocks of an 8-row unit 8 indices apart: same XCD, one after the other) ----
    int slot, row0;
    p1_index_to_rows<N, kWgRows / 4>(index, slot, row0);
    const int launch_slot = g.first1 + slot;
    const CascadeFrame cf = pair_frame(g, launch_slot);
    fetch_arguments(buf, cf);
    pass1c_split_item<N, 4, kAuxDefault, kAuxDefault>(buf, cf, g.time1[launch_slot], g.tbase1 + slot, row0, g.fault, lds, lds + SG::TW,
                                                      reinterpret_cast<int *>(lds + SG::kLdsCplx), (int)threadIdx.x, [&](int k, float keep) { ws.at(k, keep); });
    ws.write(stamps, PG::kThreads / 64, 1000ull + (unsigned long long)row0);
}

// ===================================================================================================
// LAYER-PARALLEL variants for small batches (up to ~1024 waves of row work: one cascade of 1024^2, four of 512^2, anything at
// 128^2 and 256^2).  There the standard kernels leave most of the chip idle and their duration is one wave's serial
// latency (four transforms back to back).  Here every (row, layer) pair gets its own lanes: 4x the waves, a quarter
// of the serial work each.  Same lane code, same results up to FMA contraction order.
// ===================================================================================================

// PASS 1, layer-parallel: grid (rows/8, 4); block y = the packed layer this block transforms.  Each of the four blocks
// of a row group loads and modulates the spectrum itself (the duplicates are L2 hits; the chip is idle anyway).
template <int N, int AUX_T = kAuxDefault>
__global__ __launch_bounds__(plan_wg_threads(N), 4) void k_pass1_lp(DeviceBuffers buf, FrameArgs args) {
    constexpr int Tn = plan_T(N), P = kP;
    __shared__ __attribute__((aligned(16))) cplx lds[plan_wg_lds_cplx(N) + plan_sync_flag_cplx(N, kWgRows)];
    cplx *tw_lds = lds;
    cplx *rows_lds = lds + plan_tw_total(N);
    const int tau = threadIdx.x;
    const int rw = (Tn >= 64) ? __builtin_amdgcn_readfirstlane(tau / Tn) : tau / Tn, t = tau % Tn;
    const uint32_t plane = (uint32_t)N * N;
    cplx *lds_row = rows_lds + rw * plan_region_cplx(N);
    int *sync_flags = reinterpret_cast<int *>(lds + plan_wg_lds_cplx(N));
    RowSync<N> rs;
    rs.attach(sync_flags, rw, (tau / 64) & 1);
    rs.watch(buf.status, args.c[0].fault);
    init_row_sync<N>(sync_flags, kWgRows);
    const int L = blockIdx.y;
    int slot, row0;
    p1_block_to_rows<N>(slot, row0);
    const CascadeFrame cf = args.c[slot];
    fetch_arguments(buf, cf);
    const int y = row0 + rw;
    const GBuf h0_c = make_gbuf(buf.h0 + (size_t)cf.cascade * plane, plane * 8u);
    const GBuf om_c = make_gbuf(buf.omega + (size_t)cf.cascade * plane, plane * 4u);
    const GBuf T_c = make_gbuf(buf.T + (size_t)slot * plane * kLayers, t_cascade_bytes(N));  // scratch: indexed by launch slot, reused by every batch
    cplx h[P];
    {
        TwPrefetch<N> twp;
        tw_fetch<N>(twp, buf.tw);
        cplx a[P], b[P];
        float om[P];
        Pass1<N>::load_raw(a, b, om, t, y, h0_c, om_c);
        tw_commit<N>(twp, tw_lds);
        Pass1<N>::modulate(h, a, b, om, cf.time);
    }
    const float dkx = (2.0f * kPi) / cf.tile_x, dky = (2.0f * kPi) / cf.tile_y;
    const float ky = (float)(y - N / 2) * dky;
    float ik[P];
    Pass1<N>::wave_numbers(ik, t, ky, dkx);
    cplx d[P];
    OW_SCHED_FENCE();
    switch (L) {  // block-uniform
        case 0: Pass1<N>::template layer_input<0>(d, h, ik, t, ky, dkx); break;
        case 1: Pass1<N>::template layer_input<1>(d, h, ik, t, ky, dkx); break;
        case 2: Pass1<N>::template layer_input<2>(d, h, ik, t, ky, dkx); break;
        default: Pass1<N>::template layer_input<3>(d, h, ik, t, ky, dkx); break;
    }
    OW_SCHED_FENCE();
    row_ifft<N>(d, t, lds_row, tw_lds, rs);
    rs.sync();
    Pass1<N>::stage_write(d, t, lds_row);
    lds_barrier();
    Pass1<N>::template stage_store<AUX_T>(tau, L, row0, rows_lds, T_c);
}

// PASS 2, layer-parallel: a block = plan_lp_rows(N) rows x 4 layers x N/16 lanes (512 threads).  Every (row, layer)
// lane group transforms its layer, leaves it in its LDS region in natural order, and after one block barrier each
// layer group finishes a quarter of the row's texels (fft_unpack.glsl for texels o = 4g .. 4g+3 of every lane).

template <int N, bool F32, int AUX_T = kAuxDefault, int AUX_O = kAuxDefault>
__global__ __launch_bounds__(plan_lp_threads(N), 2) void k_pass2_lp(DeviceBuffers buf, FrameArgs args) {
    constexpr int Tn = plan_T(N), P = kP, ROWS = plan_lp_rows(N), PER_LAYER = ROWS * Tn;
    __shared__ __attribute__((aligned(16))) cplx lds[plan_lp_lds_cplx(N) + plan_sync_flag_cplx(N, plan_lp_rows(N) * kLayers)];
    cplx *tw_lds = lds;
    cplx *rows_lds = lds + plan_tw_total(N);
    const int tau = threadIdx.x;
    const int g = __builtin_amdgcn_readfirstlane(tau / PER_LAYER);  // layer of this lane group (wave-uniform: PER_LAYER >= 128)
    const int r = (tau % PER_LAYER) / Tn, t = tau % Tn;
    const uint32_t plane = (uint32_t)N * N;
    constexpr int BPC = N / ROWS;
    const int slot = blockIdx.x / BPC, row0 = (blockIdx.x % BPC) * ROWS;
    const CascadeFrame cf = args.c[slot];
    fetch_arguments(buf, cf);
    const int xp = row0 + r;
    const uint32_t tex = (uint32_t)(xp * N + t);
    const GBuf T_c = make_gbuf(buf.T + (size_t)slot * plane * kLayers, t_cascade_bytes(N));  // scratch: indexed by launch slot, reused by every batch
    const GBuf disp_c = make_gbuf(buf.disp + (size_t)cf.cascade * plane, plane * 8u);
    const GBuf norm_c = make_gbuf(buf.norm + (size_t)cf.cascade * plane, plane * 8u);
    const GBuf foam_c = make_gbuf(buf.foam + (size_t)cf.cascade * plane, plane * 2u);
    const GBuf f32_c = make_gbuf(F32 ? buf.f32 + (size_t)cf.cascade * plane * 8 : nullptr, F32 ? plane * 32u : 0u);
    auto region = [&](int row, int layer) { return rows_lds + (layer * ROWS + row) * plan_region_cplx(N); };
    int *sync_flags = reinterpret_cast<int *>(lds + plan_lp_lds_cplx(N));
    RowSync<N> rs;
    rs.attach(sync_flags, g * ROWS + r, (tau / 64) & 1);
    rs.watch(buf.status, args.c[0].fault);
    init_row_sync<N>(sync_flags, ROWS * kLayers);

    TwPrefetch<N> twp;
    tw_fetch<N>(twp, buf.tw);
    cplx d[P];
    Pass2<N>::template load_layer<AUX_T>(d, t, xp, g, T_c);
    // this group's quarter of the lane's foam values (o = 4g .. 4g+3): 8 bytes of the lane's 32
    const cplx foam_bits = gload8(foam_c, Pass2<N>::foam_index(xp, t) * 2u, (uint32_t)g * 8u);
    tw_commit<N>(twp, tw_lds);
    row_ifft<N>(d, t, region(r, g), tw_lds, rs);
    rs.sync();
    {
        cplx *mine = region(r, g);
#pragma unroll
        for (int o = 0; o < P; ++o) mine[t + Tn * o] = d[OutMap<N>::slot_of(o)];
    }
    lds_barrier();
    const float fb0 = foam_bits.x, fb1 = foam_bits.y;
    const uint32_t fpk[2] = {__builtin_bit_cast(uint32_t, fb0), __builtin_bit_cast(uint32_t, fb1)};
    uint32_t fnew[2] = {0u, 0u};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int o = 4 * g + q;  // group-uniform
        const int e = t + Tn * o;
        const cplx l0 = lds_read(region(r, 0) + e), l1 = lds_read(region(r, 1) + e), l2 = lds_read(region(r, 2) + e), l3 = lds_read(region(r, 3) + e);
        const uint16_t prev = (uint16_t)((fpk[q / 2] >> (16 * (q & 1))) & 0xFFFFu);
        const uint32_t fh = Pass2<N>::template unpack_texel<F32, AUX_O>(l0, l1, l2, l3, prev, t, xp, o, tex, cf, disp_c, norm_c, f32_c);
        fnew[q / 2] |= fh << (16 * (q & 1));
    }
    const float fn0 = __builtin_bit_cast(float, fnew[0]), fn1 = __builtin_bit_cast(float, fnew[1]);
    gstore8(foam_c, Pass2<N>::foam_index(xp, t) * 2u, (uint32_t)g * 8u, cplx{fn0, fn1});
}

// ===================================================================================================
// LAYER-PARALLEL + COMPACT INTERMEDIATE (N >= 256): the small-batch kernels on the two-and-a-half-layer intermediate.
// Pass 1: one ITEM = (8 rows, L): L = 0..2 the compact layers (layer 1 exists for the upper half of the rows only), L = 3..5 the
// three extra transforms of texel row 0 (rows 0..7 of a cascade only).  Pass 2: one item = plan_lp_rows(N) rows; the four lane
// groups of a row compute F0..F3.  The item bodies are functions of (item, lane) so that they can be driven by other launch shapes
// (the stand-alone kernels below: one item per block; tools/tick_loop_experiment.h: a persistent loop over items and ticks).
// ===================================================================================================
struct NoStamps {
    __device__ __forceinline__ void at(int, float) {}
};
// issued(a15, om15): called right after the item's global loads have been issued (the stand-alone kernel commits its
// twiddle prefetch there); tslot = slot of the scratch intermediate / side buffers this item writes
template <int N, int AUX_T, class Issued, class WS>
__device__ __forceinline__ void pass1c_lp_item(const DeviceBuffers &buf, const CascadeFrame &cf, float time, int tslot, int row0, int L, int tau,
                                               cplx *tw_lds, cplx *rows_lds, RowSync<N> &rs, Issued issued, WS &ws) {
    constexpr int Tn = plan_T(N), P = kP;
    const int rw = (Tn >= 64) ? __builtin_amdgcn_readfirstlane(tau / Tn) : tau / Tn, t = tau % Tn;
    const uint32_t plane = (uint32_t)N * N;
    cplx *lds_row = rows_lds + rw * plan_region_cplx(N);
    const int y = row0 + rw;
    const GBuf h0_c = make_gbuf(buf.h0 + (size_t)cf.cascade * plane, plane * 8u);
    const GBuf om_c = make_gbuf(buf.omega + (size_t)cf.cascade * plane, plane * 4u);
    const GBuf T_c = make_gbuf(buf.T + (size_t)tslot * plane * kLayers, t_cascade_bytes(N));
    const GBuf pcol_c = make_gbuf(buf.pcol + (size_t)tslot * N, (uint32_t)N * 8u);
    const GBuf rrow_c = make_gbuf(buf.rrow + (size_t)tslot * N * 4, (uint32_t)N * 32u);
    cplx h[P];
    {
        cplx a[P], b[P];
        float om[P];
        Pass1<N>::load_raw(a, b, om, t, y, h0_c, om_c);
        ws.at(1, 0.0f);                 // loads issued
        issued();
        ws.at(3, a[15].x + om[15]);     // the wave's own data has arrived
        Pass1<N>::modulate(h, a, b, om, time);
    }
    ws.at(4, h[15].x);                  // modulated
    const float dkx = (2.0f * kPi) / cf.tile_x, dky = (2.0f * kPi) / cf.tile_y;
    const float ky = (float)(y - N / 2) * dky;
    float ik[P];
    Pass1<N>::wave_numbers(ik, t, ky, dkx);
    if (L == 0 && t == 0) gstore8<AUX_T>(pcol_c, Pass2<N>::pcol_index(y) * 8u, 0u, Pass1<N>::column_term(h, ik, t, dkx));
    cplx d[P];
    OW_SCHED_FENCE();
    switch (L) {  // item-uniform
        case 0: Pass1<N>::template layer_input_c<0>(d, h, ik, t, ky, dkx); break;
        case 1: Pass1<N>::template layer_input_c<1>(d, h, ik, t, ky, dkx); break;
        case 2: Pass1<N>::template layer_input_c<2>(d, h, ik, t, ky, dkx); break;
        case 3: Pass1<N>::template row0_input<1>(d, h, ik, t, ky, dkx); break;
        case 4: Pass1<N>::template row0_input<2>(d, h, ik, t, ky, dkx); break;
        default: Pass1<N>::template row0_input<3>(d, h, ik, t, ky, dkx); break;
    }
    OW_SCHED_FENCE();
    ws.at(5, d[15].x);                  // layer input built
    row_ifft<N>(d, t, lds_row, tw_lds, rs);
    ws.at(6, d[15].x);                  // transformed
    if (L >= 3) {  // straight to the side buffer, lanes of row 0 only (the other rows of the item computed nothing of use)
        if (y == 0) {
#pragma unroll
            for (int o = 0; o < P; ++o) gstore8<AUX_T>(rrow_c, (uint32_t)(t + Tn * o) * 32u, (uint32_t)(L - 2) * 8u, d[OutMap<N>::slot_of(o)]);
        }
        return;
    }
    rs.sync();
    Pass1<N>::stage_write(d, t, lds_row);
    lds_barrier();                      // (block-wide: every item of a block takes this path or none does)
    ws.at(7, 0.0f);                     // staged, block barrier passed
    Pass1<N>::template stage_store<AUX_T>(tau, L, row0, rows_lds, T_c);
    ws.at(8, 0.0f);                     // stores issued
}

template <int N, int AUX_T = kAuxDefault, bool STAMPS = false>
__global__ __launch_bounds__(plan_wg_threads(N), 4) void k_pass1c_lp(DeviceBuffers buf, FrameArgs args, Stamp *stamps = nullptr) {
    WaveStamps<STAMPS> ws;
    ws.at(0, 0.0f);
    static_assert(plan_T(N) >= 16, "Pass2::load_c1 needs N/16 to be a multiple of the 16-row line");
    int slot, row0;
    p1_block_to_rows<N>(slot, row0);
    const int L = blockIdx.y;                 // 0..2: layer, 3..5: row-0 transform Q = L - 2
    if (L == 1 && row0 < N / 2) return;       // hz of the lower rows is the conjugate of the mirrored rows': not transformed
    if (L >= 3 && row0 != 0) return;
    __shared__ __attribute__((aligned(16))) cplx lds[plan_wg_lds_cplx(N) + plan_sync_flag_cplx(N, kWgRows)];
    cplx *tw_lds = lds;
    cplx *rows_lds = lds + plan_tw_total(N);
    const int tau = threadIdx.x;
    int *sync_flags = reinterpret_cast<int *>(lds + plan_wg_lds_cplx(N));
    RowSync<N> rs;
    rs.attach(sync_flags, tau / plan_T(N), (tau / 64) & 1);
    rs.watch(buf.status, args.c[0].fault);
    init_row_sync<N>(sync_flags, kWgRows);
    TwPrefetch<N> twp;
    tw_fetch<N>(twp, buf.tw);
    const CascadeFrame cf = args.c[slot];
    fetch_arguments(buf, cf);
    pass1c_lp_item<N, AUX_T>(buf, cf, cf.time, slot, row0, L, tau, tw_lds, rows_lds, rs, [&] {
        tw_commit<N>(twp, tw_lds);
        ws.at(2, tw_lds[0].x);          // table in LDS, block barrier passed
    }, ws);
    ws.write(stamps, (plan_wg_threads(N) + 63) / 64, (unsigned long long)L);   // [14] = stores acknowledged
}

// tau = thread index inside the item's plan_lp_threads(N) lanes; rows_lds = the item's plan_lp_rows(N) x 4 row regions
// foam: the lane's four FP16 foam values (8 bytes).  foam_io & 1: load them from the foam plane first; & 2: store them back at the end
// (a caller that runs consecutive ticks of the same rows keeps them in registers in between)
// The layer-parallel pass 2 of RH consecutive columns in two halves, as functions of the lane index tau (0 .. 4 RH N/16 - 1):
//   front: the lane group's transform -- loads (+ the lane's foam values when foam_load), row IFFT, results staged in `regions`
//   back : (after a block barrier) all four transforms of a texel from `regions` -> unpack, map stores; foam_bits in -> out
// pass2c_lp_item below runs them back to back; the pipelined tick groups run front(tick j + 1) beside back(tick j).
template <int N, int RH, int AUX_T, class Issued, class WS>
__device__ __forceinline__ void pass2c_lp_front(const DeviceBuffers &buf, const CascadeFrame &cf, int tslot, int row0, int tau, cplx *tw_lds, cplx *regions,
                                                RowSync<N> &rs, Issued issued, WS &ws, cplx &foam_bits, bool foam_load) {
    constexpr int Tn = plan_T(N), P = kP, PER_LAYER = RH * Tn;
    const int g = __builtin_amdgcn_readfirstlane(tau / PER_LAYER);  // which of F0..F3 this lane group transforms (wave-uniform)
    const int r = (tau % PER_LAYER) / Tn, t = tau % Tn;
    const uint32_t plane = (uint32_t)N * N;
    const int xp = row0 + r;
    const GBuf T_c = make_gbuf(buf.T + (size_t)tslot * plane * kLayers, t_cascade_bytes(N));
    const GBuf pcol_c = make_gbuf(buf.pcol + (size_t)tslot * N, (uint32_t)N * 8u);
    const GBuf rrow_c = make_gbuf(buf.rrow + (size_t)tslot * N * 4, (uint32_t)N * 32u);
    const GBuf foam_c = make_gbuf(buf.foam + (size_t)cf.cascade * plane, plane * 2u);
    const float dky = (2.0f * kPi) / cf.tile_y;
    cplx *mine = regions + (g * RH + r) * plan_region_cplx(N);

    cplx d[P];
    switch (g) {  // group-uniform
        case 0: Pass2<N>::template load_layer<AUX_T>(d, t, xp, 0, T_c); break;
        case 1:
            Pass2<N>::template load_layer<AUX_T>(d, t, xp, 0, T_c);
            Pass2<N>::template derive_dx<AUX_T>(d, t, xp, dky, pcol_c);
            break;
        case 2: Pass2<N>::template load_c1<AUX_T>(d, t, xp, dky, T_c); break;
        default: Pass2<N>::template load_layer<AUX_T>(d, t, xp, 2, T_c); break;
    }
    if (g != 0) Pass2<N>::put_row0(d, t, gload8<AUX_T>(rrow_c, (uint32_t)xp * 32u, (uint32_t)g * 8u));
    if (foam_load) foam_bits = gload8(foam_c, Pass2<N>::foam_index(xp, t) * 2u, (uint32_t)g * 8u);
    ws.at(1, 0.0f);                     // loads issued
    issued();
    ws.at(3, d[15].x + foam_bits.x);    // the wave's own data has arrived
    row_ifft<N>(d, t, mine, tw_lds, rs);
    ws.at(6, d[15].x);                  // transformed
    rs.sync();
#pragma unroll
    for (int o = 0; o < P; ++o) mine[t + Tn * o] = d[OutMap<N>::slot_of(o)];
}
// the output side of one cascade (built once per block: the scalar loads and address arithmetic behind them stay off the per-tick path)
struct P2Out {
    GBuf disp_c, norm_c, foam_c, f32_c;
};
template <int N, bool F32>
__device__ __forceinline__ P2Out make_p2_out(const DeviceBuffers &buf, const CascadeFrame &cf) {
    const uint32_t plane = (uint32_t)N * N;
    return P2Out{make_gbuf(buf.disp + (size_t)cf.cascade * plane, plane * 8u), make_gbuf(buf.norm + (size_t)cf.cascade * plane, plane * 8u),
                 make_gbuf(buf.foam + (size_t)cf.cascade * plane, plane * 2u),
                 make_gbuf(F32 ? buf.f32 + (size_t)cf.cascade * plane * 8 : nullptr, F32 ? plane * 32u : 0u)};
}
template <int N, bool F32, int RH, int AUX_O, class WS>
__device__ __forceinline__ void pass2c_lp_back(const P2Out &out, const CascadeFrame &cf, int row0, int tau, const cplx *regions, WS &ws, cplx &foam_bits,
                                               bool foam_store) {
    constexpr int Tn = plan_T(N), PER_LAYER = RH * Tn;
    const int g = __builtin_amdgcn_readfirstlane(tau / PER_LAYER);
    const int r = (tau % PER_LAYER) / Tn, t = tau % Tn;
    const int xp = row0 + r;
    const uint32_t tex = (uint32_t)(xp * N + t);
    const GBuf disp_c = out.disp_c, norm_c = out.norm_c, foam_c = out.foam_c, f32_c = out.f32_c;
    auto region = [&](int row, int layer) { return regions + (layer * RH + row) * plan_region_cplx(N); };
    const float fb0 = foam_bits.x, fb1 = foam_bits.y;
    const uint32_t fpk[2] = {__builtin_bit_cast(uint32_t, fb0), __builtin_bit_cast(uint32_t, fb1)};
    uint32_t fnew[2] = {0u, 0u};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int o = 4 * g + q;  // group-uniform
        const int e = t + Tn * o;
        const cplx f0 = lds_read(region(r, 0) + e), f1 = lds_read(region(r, 1) + e), f2 = lds_read(region(r, 2) + e), f3 = lds_read(region(r, 3) + e);
        // back to the reference's packing: (hx, hy), (hz, dhy_dx), (dhy_dz, dhx_dx), (dhz_dz, dhz_dx)
        const cplx l1 = cplx{f2.x, f1.y}, l2 = cplx{f3.x, f1.x}, l3 = cplx{f3.y, f2.y};
        const uint16_t prev = (uint16_t)((fpk[q / 2] >> (16 * (q & 1))) & 0xFFFFu);
        const uint32_t fh = Pass2<N>::template unpack_texel<F32, AUX_O>(f0, l1, l2, l3, prev, t, xp, o, tex, cf, disp_c, norm_c, f32_c);
        fnew[q / 2] |= fh << (16 * (q & 1));
    }
    const float fn0 = __builtin_bit_cast(float, fnew[0]), fn1 = __builtin_bit_cast(float, fnew[1]);
    foam_bits = cplx{fn0, fn1};
    if (foam_store) gstore8(foam_c, Pass2<N>::foam_index(xp, t) * 2u, (uint32_t)g * 8u, foam_bits);
    ws.at(8, 0.0f);                     // unpacked, stores issued
}
template <int N, bool F32, int AUX_T, int AUX_O, class Issued, class WS>
__device__ __forceinline__ void pass2c_lp_item(const DeviceBuffers &buf, const CascadeFrame &cf, int tslot, int row0, int tau, cplx *tw_lds,
                                               cplx *rows_lds, RowSync<N> &rs, Issued issued, WS &ws, cplx &foam_bits, int foam_io = 3) {
    constexpr int ROWS = plan_lp_rows(N);
    const P2Out out = make_p2_out<N, F32>(buf, cf);  // (before the loads: its scalar work runs under their latency)
    pass2c_lp_front<N, ROWS, AUX_T>(buf, cf, tslot, row0, tau, tw_lds, rows_lds, rs, issued, ws, foam_bits, (foam_io & 1) != 0);
    lds_barrier();
    ws.at(7, 0.0f);                     // all four transforms of the row are in LDS
    pass2c_lp_back<N, F32, ROWS, AUX_O>(out, cf, row0, tau, rows_lds, ws, foam_bits, (foam_io & 2) != 0);
}

template <int N, bool F32, int AUX_T = kAuxDefault, int AUX_O = kAuxDefault, bool STAMPS = false>
__global__ __launch_bounds__(plan_lp_threads(N), 2) void k_pass2c_lp(DeviceBuffers buf, FrameArgs args, Stamp *stamps = nullptr) {
    WaveStamps<STAMPS> ws;
    ws.at(0, 0.0f);
    constexpr int Tn = plan_T(N), ROWS = plan_lp_rows(N), PER_LAYER = ROWS * Tn;
    static_assert(Tn >= 16, "Pass2::load_c1 needs N/16 to be a multiple of the 16-row line");
    __shared__ __attribute__((aligned(16))) cplx lds[plan_lp_lds_cplx(N) + plan_sync_flag_cplx(N, plan_lp_rows(N) * kLayers)];
    cplx *tw_lds = lds;
    cplx *rows_lds = lds + plan_tw_total(N);
    const int tau = threadIdx.x;
    constexpr int BPC = N / ROWS;
    const int slot = blockIdx.x / BPC, row0 = (blockIdx.x % BPC) * ROWS;
    int *sync_flags = reinterpret_cast<int *>(lds + plan_lp_lds_cplx(N));
    RowSync<N> rs;
    rs.attach(sync_flags, (tau / PER_LAYER) * ROWS + (tau % PER_LAYER) / Tn, (tau / 64) & 1);
    rs.watch(buf.status, args.c[0].fault);
    init_row_sync<N>(sync_flags, ROWS * kLayers);
    const CascadeFrame cf = args.c[slot];
    fetch_arguments(buf, cf);
    TwPrefetch<N> twp;
    tw_fetch<N>(twp, buf.tw);
    cplx foam_bits;
    pass2c_lp_item<N, F32, AUX_T, AUX_O>(buf, cf, slot, row0, tau, tw_lds, rows_lds, rs, [&] {
        tw_commit<N>(twp, tw_lds);
        ws.at(2, tw_lds[0].x);          // table in LDS, block barrier passed
    }, ws, foam_bits);
    ws.write(stamps, (plan_lp_threads(N) + 63) / 64, (unsigned long long)(tau / PER_LAYER));   // [14] = stores acknowledged
}

// ===================================================================================================
// TICK GROUPS for ow_run on small batches (the layer-parallel compact family).  Pass 1 of a tick depends on the spectrum and the
// time only, never on earlier results; pass 2 of a tick depends on its own pass 1 and, through the foam recurrence, on pass 2 of
// the previous tick OF THE SAME ROWS.  So a run of ticks is launched in groups of D:
//     [pass 1 of group 0]  [pass 2 of group 0 + pass 1 of group 1]  ...  [pass 2 of the last group]
// where, inside one launch, a pass-2 block walks through the D ticks of its rows one after the other (foam stays in registers in
// between) while other blocks do pass 1 of the next D ticks side by side.  K ticks cost K / D + 1 launches instead of 2 K, the
// ~3 us of launch / first-fetch latency are paid once per D ticks, and a chip that one small tick cannot fill (256^2 x 4: one wave
// per SIMD at best) is filled by D of them.  Nothing crosses blocks inside a launch -- the intermediates a launch reads were
// finished by the previous launch -- and the scratch intermediate (T, pcol, rrow) holds 2 D ticks (tick t lives in slots
// (t mod 2D) * count ...), so pass 1 writes where pass 2 of the launch before last read.  Per texel the arithmetic and its order are
// those of the one-launch-per-pass kernels: results are bit-identical (tests/test_tick_groups.py).
// Blocks [0, n2) are pass-2 items (each loops over d2 ticks), blocks [n2, n2 + d1 * n1) pass-1 items of d1 ticks (TickPlan: Q
// side-by-side 8-row items per block -- all of the same layer in the layer-parallel form, each doing all its layers in k_pass1c's
// form, g.p1_compact, which the runtime picks for all but the smallest ticks); d2 or d1 may be 0 (the two ends of a run).
// ===================================================================================================
template <int N, bool F32, bool STAMPS = false, bool PIPE = false>
__global__ __launch_bounds__(plan_lp_threads(N), 4) void k_tick_group_c_lp(DeviceBuffers buf, FrameArgs args, TickGroupArgs g, Stamp *stamps = nullptr) {
    using TP = TickPlan<N>;
    constexpr int ROWS = plan_lp_rows(N), SUB = plan_wg_threads(N);
    static_assert(!plan_row_spans_waves(N), "small-batch sizes only (N <= 1024)");
    __shared__ __attribute__((aligned(16))) cplx lds[plan_lp_lds_cplx(N) + (PIPE ? plan_lp_handoff_cplx(N) : 0)];
    cplx *tw_lds = lds;
    cplx *rows_lds = lds + plan_tw_total(N);
    RowSync<N> rs;
    NoStamps ws;
    WaveStamps<STAMPS> tl;  // developer timeline (tools/kbench_small): pass 2 -- [j] = start of tick j, [d2] = end; pass 1 -- [0], [1]
    tl.at(0, 0.0f);
    TwPrefetch<N> twp;
    tw_fetch<N>(twp, buf.tw);
    // (PIPE is its own instantiation, not a run-time branch: with one wave per SIMD the instruction stream of a small launch is
    //  latency bound, and the second copy of the pass-2 code in the kernel cost the plain form 1.5 us per tick at 256^2)
    if constexpr (PIPE && ROWS >= 2) {
        if ((int)blockIdx.x < g.n2) {
            // ---- pass 2, pipelined: the block's two halves (4 lane groups x RH columns each) take alternate ticks of the same RH columns.
            // In step s one half runs the FRONT of tick s (loads, transform, results staged in its own regions) while the other runs
            // the BACK of tick s - 1 (unpack from its regions, map stores); the foam values -- all that one tick hands to the next --
            // cross from half to half through LDS.  A tick then costs max(front, back) on the chain of ticks instead of their sum:
            // with one wave per SIMD (a 256^2 x 4 launch has 1024 pass-2 waves) that chain IS the launch.
            constexpr int RH = TP::kPipeRows, HALF = plan_lp_threads(N) / 2;
            const int half = __builtin_amdgcn_readfirstlane((int)threadIdx.x / HALF);
            const int item = blockIdx.x, slot = item / (N / RH), row0 = (item % (N / RH)) * RH;
            const CascadeFrame cf = args.c[slot];
            fetch_arguments(buf, cf);
            cplx *regions = rows_lds + half * (kLayers * RH) * plan_region_cplx(N);
            cplx *handoff = lds + plan_lp_lds_cplx(N);
            const P2Out out = make_p2_out<N, F32>(buf, cf);
            tw_commit<N>(twp, tw_lds);
            cplx foam_bits = cplx{0.0f, 0.0f};
            for (int s = 0; s <= g.d2; ++s) {
                const int tau = opaque((int)threadIdx.x % HALF);
                if (s < g.d2 && (s & 1) == half)
                    pass2c_lp_front<N, RH, kAuxDefault>(buf, cf, g.tbase2[s] + slot, row0, tau, tw_lds, regions, rs, [] {}, ws, foam_bits, s == 0);
                if (s >= 1 && ((s - 1) & 1) == half) {
                    if (s > 1) foam_bits = lds_read(handoff + tau);  // (written by the other half one step ago)
                    pass2c_lp_back<N, F32, RH, kAuxDefault>(out, cf, row0, tau, regions, ws, foam_bits, s == g.d2);
                    handoff[tau] = foam_bits;
                }
                lds_barrier();
                if (s + 1 <= 13) tl.at(s + 1, foam_bits.x);
            }
            tl.write(stamps, plan_lp_threads(N) / 64, 0ull);
            return;
        }
    }
    // (pass-2 blocks first, then the pass-1 blocks.  Alternating the two kinds in chunks of 8, as the tick-pair kernels do, was measured for the
    //  look-ahead's launches of one tick of pass 2 beside pass 1 of later ones -- and LOSES 3 - 9 %: 1024^2 x 4 on the reference's schedule 84.1 -> 87.0 us
    //  per update, 1024^2 x 1 tick by tick 19.9 -> 21.8, 512^2 x 4 19.8 -> 21.5; profiles/EXPERIMENTS.md round 6)
    if (!PIPE && (int)blockIdx.x < g.n2) {  // ---- pass 2 of d2 consecutive ticks of the same rows (block-uniform branch) ----
        const int item = blockIdx.x;
        const int slot = item / (N / ROWS), row0 = (item % (N / ROWS)) * ROWS;
        const CascadeFrame cf = args.c[slot];  // (pass 2 does not use the time)
        fetch_arguments(buf, cf);
        cplx foam_bits;
        for (int j = 0; j < g.d2; ++j) {
            if (j > 0 && j <= 13) tl.at(j, foam_bits.x);  // (slots 14, 15 of the stamp record are write()'s: end time and tag)
            const int tau = opaque((int)threadIdx.x);  // per tick: lane-derived offsets are recomputed, not carried around the loop
            const int io = (j == 0 ? 1 : 0) | (j == g.d2 - 1 ? 2 : 0);
            if (j == 0) {
                pass2c_lp_item<N, F32, kAuxDefault, kAuxDefault>(buf, cf, g.tbase2[0] + slot, row0, tau, tw_lds, rows_lds, rs, [&] { tw_commit<N>(twp, tw_lds); }, ws,
                                                                 foam_bits, io);
            } else {
                lds_barrier();  // the previous tick's LDS reads are done
                pass2c_lp_item<N, F32, kAuxDefault, kAuxDefault>(buf, cf, g.tbase2[j] + slot, row0, tau, tw_lds, rows_lds, rs, [] {}, ws, foam_bits, io);
            }
        }
        tl.at(g.d2 < 13 ? g.d2 : 13, foam_bits.x);
        tl.write(stamps, plan_lp_threads(N) / 64, 0ull);
        return;
    }
    // ---- pass 1: tick j of the later group ----
    const int b1 = (int)blockIdx.x - g.n2, j = b1 / g.n1, item = b1 % g.n1;
    const int tau = threadIdx.x;
    const int sub = __builtin_amdgcn_readfirstlane(tau / SUB), tau_sub = tau % SUB;
    if (g.p1_compact) {  // (launch-uniform) Q 8-row items side by side, each doing all its layers (k_pass1c's body): no redundant modulation
        int slot, row0;
        TP::decode_compact(item, sub, slot, row0);
        const CascadeFrame cf = args.c[g.first1 + j * g.step1 + slot];  // (first1, step1: the pass-1 cascades may sit in launch slots of their own, ow_runtime.hip lookahead_launch)
        fetch_arguments(buf, cf);
        pass1c_item<N, kAuxDefault, kAuxDefault>(buf, cf, g.time1[j][g.first1 + slot], g.tbase1[j] + slot, row0, tau_sub, tw_lds,
                                                 rows_lds + sub * kWgRows * plan_region_cplx(N), rs, [&] { tw_commit<N>(twp, tw_lds); }, [](int, float) {});
        tl.at(1, 0.0f);
        tl.write(stamps, plan_lp_threads(N) / 64, 1000ull + (unsigned long long)j);
        return;
    }
    // Q items of one layer side by side (the layer-parallel form)
    int L, slot, row0;
    const bool active = TP::decode(item, sub, g.slots, L, slot, row0);
    // (an idle sub-block of a row-0 item still takes part in the table's block barrier; the row-0 path has no other block barrier,
    //  and in the layer paths every sub-block of the block is active)
    const CascadeFrame cf = args.c[g.first1 + j * g.step1 + slot];
    fetch_arguments(buf, cf);
    if (active) {
        pass1c_lp_item<N, kAuxDefault>(buf, cf, g.time1[j][g.first1 + slot], g.tbase1[j] + slot, row0, L, tau_sub, tw_lds, rows_lds + sub * kWgRows * plan_region_cplx(N),
                                       rs, [&] { tw_commit<N>(twp, tw_lds); }, ws);
    } else {
        tw_commit<N>(twp, tw_lds);
    }
    tl.at(1, 0.0f);
    tl.write(stamps, plan_lp_threads(N) / 64, 1000ull + (unsigned long long)j);
}

// TICK PAIRS on the compact family (ow_run on the batches that family serves): the ticks of a run are a stream of batches of at
// most 4 Mi texels (a tick of more is two batches), and ONE launch does pass 2 of one batch (k_pass2c's blocks: 8 columns) and pass 1 of
// the NEXT batch of the stream (k_pass1c's blocks: 8 rows; the same cascades one tick later, or the tick's other cascades) --
// independent work; the scratch intermediate is two batches deep (g.tbase2[0] / g.tbase1[0]).  Chunks of 8 blocks (one block per XCD,
// so both block -> rows maps keep their XCD placement) alternate between the two passes: a CU holds blocks of both, pass 1's exposed
// transform time overlaps pass 2's memory time, and the launch gap and the tail of one kernel per batch are gone.  Same item bodies
// as k_pass1c / k_pass2c: results are bit-identical to one launch per pass.  g.slots2 / g.slots1 = 0 at the two ends of a run.
// (Measured, MI355X, us per tick against k_pass1c + k_pass2c: 1024^2 x 2 27.6 / 38.5, x 3 42.4 / 50.6, x 4 53.4 / 57.1, 512^2 x 8 27.1 / 34.8.
//  Deeper groups -- a block walking through 2 or 4 ticks of its columns as in k_tick_group_c_lp -- gain nothing more here and lose
//  once the deeper scratch leaves the Infinity Cache (profiles/r02_tick_pairs_compact.txt).  N <= 1024: at 2048^2 the two passes'
//  blocks are of another shape, k_tick_pair_c_split above.)
template <int N, bool F32>
__global__ __launch_bounds__(plan_wg_threads(N), 4) void k_tick_pair_c(DeviceBuffers buf, PairArgs g) {
    static_assert(!plan_row_spans_waves(N), "N <= 1024");
    __shared__ __attribute__((aligned(16))) cplx lds[plan_wg_lds_cplx(N)];
    cplx *tw_lds = lds;
    cplx *rows_lds = lds + plan_tw_total(N);
    RowSync<N> rs;
    TwPrefetch<N> twp;
    tw_fetch<N>(twp, buf.tw);
    // g.n2 pass-2 blocks and g.n1 pass-1 blocks (multiples of 8, either may be 0): alternate in chunks of 8 while both last
    int index = blockIdx.x;
    bool first;  // is this a pass-1 block?
    {
        const int both = 2 * (g.n2 < g.n1 ? g.n2 : g.n1), chunk = index >> 3;
        if (index < both) {
            first = chunk & 1;
            index = ((chunk >> 1) << 3) + (index & 7);
        } else {
            first = g.n2 < g.n1;
            index -= both / 2;
        }
    }
    int slot, row0;
    if (!first) {
        constexpr int BPC = N / kWgRows;
        slot = index / BPC;
        row0 = (index % BPC) * kWgRows;
    } else {
        p1_index_to_rows<N>(index, slot, row0);
    }
    const int launch_slot = (first ? g.first1 : g.first2) + slot;  // slot: index inside the batch = scratch slot of its intermediate
    const CascadeFrame cf = pair_frame(g, launch_slot);
    fetch_arguments(buf, cf);
    if (!first) {
        uint32_t foam_pk[kP / 2];
        pass2c_item<N, F32, kAuxDefault, kAuxNT>(buf, cf, g.tbase2 + slot, row0, (int)threadIdx.x, tw_lds, rows_lds, rs, [&] { tw_commit<N>(twp, tw_lds); }, foam_pk);
    } else {
        pass1c_item<N, kAuxDefault, kAuxDefault>(buf, cf, g.time1[launch_slot], g.tbase1 + slot, row0, (int)threadIdx.x, tw_lds, rows_lds, rs,
                                                 [&] { tw_commit<N>(twp, tw_lds); }, [](int, float) {});
    }
}

}  // namespace ow
