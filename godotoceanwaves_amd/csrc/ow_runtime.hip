// ow_runtime.hip -- host runtime behind include/ocean_waves.h: the WaveGenerator of
// assets/water/wave_generator.gd re-expressed as a HIP stream program.
//
//   init_gpu (:17-54)  -> ow_create : one allocation per resource for all array layers
//   update   (:90-109) -> ow_update : flush leftovers, advance time / foam rates, arm cascades
//   _process (:56-63)  -> ow_process: one armed cascade, highest index first
//   _update  (:65-85)  -> enqueue() : [spectrum kernel if dirty] + pass-1 + pass-2 for a batch of cascades
//
// The reference issues six dispatches per cascade; here a batch of cascades is two launches.
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "ow_internal.h"
#include "ow_kernels.h"
#include "ow_tables.h"

namespace {
thread_local std::string g_last_error;
}

namespace ow {
ow_status fail(ow_status st, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return st;
}
void set_last_error(const char *message) { g_last_error = message ? message : ""; }
}  // namespace ow

namespace {
using ow::fail;
constexpr double kHostG = 9.81;  // wave_generator.gd:5
}  // namespace

struct ow_context {
    int n = 0, cascades = 0, layers = 0, device = 0;
    float depth = 20.0f;
    int kernel_mode = 0;  // 0 = by batch size, 1 = standard, 2 = layer-parallel, 3 = compact-intermediate kernels (OW_FLAG_KERNELS_*)
    int last_family = 0;  // kernel family of the most recent batch
    hipStream_t stream = nullptr;
    bool own_stream = false, own_disp = false, own_norm = false;
    // TWO CHAINS (ow_kernels.h): tick-pair launches of four 1024^2 cascades a side go out as two launches of two cascades, the second halves on side_stream.
    // side_active: work of the second chain is in flight that `stream` has not been made to wait for -- since the fork NOTHING but first-chain launches has
    // been enqueued on `stream` (everything else goes through main_stream(), which joins first).
    hipStream_t side_stream = nullptr;
    hipEvent_t side_fork_ev = nullptr, side_join_ev = nullptr;
    bool side_active = false;
    uint64_t split_launches = 0;  // launches that went out as two chains (ow_chain_stats)
    ow::DeviceBuffers buf{};
    ow::cplx *tw_dev = nullptr, *tw_split_dev = nullptr, *tw_half_dev = nullptr;
    // generator state per invocation of update() (wave_generator.gd:13-15).  The reference keeps a reference to the caller's
    // Array; a C caller's memory is only borrowed for the duration of a call, so the context keeps COPIES of the armed records
    // (ow_set_cascade_params / ow_get_cascade_params are the explicit form of "the parameter objects are live")
    ow_cascade_params pass_parameters[OW_MAX_CASCADES] = {};
    int pass_count = 0;
    int pass_num_cascades_remaining = 0;
    // device status word: page-locked host memory mapped into the device; kernels OR error bits into it (a bounded
    // device-side spin that gave up), every synchronising entry point turns a non-zero word into OW_ERR_HIP
    uint32_t *status_host = nullptr;
    uint32_t inject_fault = 0;  // ow_debug_inject_fault: applied to the next batch only
    // The status word is consumed by the first synchronising call that sees it; the failure itself is sticky: until the next batch
    // is enqueued every call that hands out map bytes (ow_get_maps, ow_get_maps_f32, ow_sample_surface) keeps failing, and so does
    // the ow_readback_wait of every layer whose copy was in flight when the word was consumed (readback_faulted).
    // Both are bit masks over the array layers: a synchronising call that finds the word set marks the layers recomputed by the batches
    // enqueued since the previous synchronisation (enqueued_since_sync); a layer's mark is lifted only by a later batch that recomputes
    // THAT layer (the reference's schedule enqueues one cascade per call: the other layers keep the faulted batch's bytes).
    uint32_t maps_faulted = 0, enqueued_since_sync = 0;
    uint32_t readback_faulted = 0;
    ow_push_constants pc_words[OW_MAX_CASCADES] = {};  // what the reference would have packed for each cascade's most recent launch (ow_get_push_constants)
    bool pc_valid[OW_MAX_CASCADES] = {};
    // pc_words[i].spectrum are the constants layer i's RESIDENT spectrum (h0, omega) was generated from -- k_spectrum is a deterministic function
    // of those thirteen words and the map size, so a dirty record that packs to the same words is served by what is there (spectrum_is_resident)
    bool spectrum_resident[OW_MAX_CASCADES] = {};
    bool always_regenerate = false;  // OW_FLAG_ALWAYS_REGENERATE_SPECTRUM: every dirty flag launches k_spectrum, as the reference does
    uint64_t spectra_generated = 0, spectra_skipped = 0;  // ow_spectrum_stats
    // ow_update_all's adaptive look-ahead (lookahead_tick below): a pass 1 of the NEXT tick, speculated with the caller's last delta
    struct Lookahead {
        static constexpr int kMaxAhead = 4;  // ticks of pass 1 one launch may compute ahead (group kernel; the pair kernel takes one)
        bool armed = false;            // the scratch holds pass 1 of `queued` ticks that nothing has disturbed since
        int count = 0, mode = 0;       // cascades per tick; 1 = compact family (pair kernel), 2 = layer-parallel compact family (group kernel)
        int queued = 0, head = 0;      // ring of ticks computed ahead: entries head, head + 1, .. (mod kMaxAhead)
        int group[kMaxAhead] = {};     // scratch group (of `stride` launch slots) that holds each entry's intermediate
        float time[kMaxAhead][OW_MAX_CASCADES] = {};  // the FP32 times each entry was computed with, per launch slot
        int cascade[kMaxAhead][OW_MAX_CASCADES] = {};  // which cascades (per launch slot of the launch that will use the entry), and
        float tile_x[kMaxAhead][OW_MAX_CASCADES] = {}, tile_y[kMaxAhead][OW_MAX_CASCADES] = {};  // the tile lengths their pass 1 was computed with
        int cur_group = 0;             // group that held the most recent launch's own intermediate
        double last_delta = -1.0;      // the previous ow_update's delta, and for how many calls in a row it has been the same
        double streak_delta = -1.0;    // ... "the same" = equal to the delta that STARTED the streak (a slowly ramping delta is not one unbroken streak)
        int streak = 0;
        int prev_run = 1;              // updates in the caller's previous run of equal deltas (1: none that says anything)
        uint64_t hits = 0, speculated = 0;
        bool hold = false;             // ow_run is about to merge the following ticks itself: its first tick must not speculate for them
        int certain = 0;               // ticks the caller GUARANTEES will follow with the same delta (ow_run's own remaining ticks): speculated without evidence
    } la;
    // ow_run after ow_run (run_impl): what the last launch of a run computed ahead for the first launch of the NEXT run like it
    struct RunAhead {
        bool armed = false;          // the scratch holds that pass 1 and nothing has disturbed it since
        int kind = 0;                // 1 = tick groups: pass 1 of `ticks` consecutive ticks of all `count` cascades; 2 = tick pairs: pass 1 of one batch, one tick
        int count = 0;               // cascades per tick of the run that computed it
        int D = 0, ticks = 0, pos = 0;  // kind 1: ticks per group of that run, ticks computed ahead, ring position (in ticks, mod 2 D) of the first of them
        int batch = 0, first = 0, size = 0, parity = 0;  // kind 2: the batch (its first launch slot, its cascades) and the half of the scratch its intermediate is in
        float time[ow::kMaxTickGroup][OW_MAX_CASCADES] = {};  // the FP32 times it was computed with, per tick (kind 2: entry 0) and launch slot
        float tile_x[OW_MAX_CASCADES] = {}, tile_y[OW_MAX_CASCADES] = {};  // ... and the tile lengths, per launch slot
        bool last_was_run = false;   // the most recent tick-advancing call was an ow_run (lowered by ow_update / ow_update_all / ow_process from outside a run)
        int last_count = 0;
        double last_delta = 0.0;
        int run_streak = 0;          // how many runs like this one (same delta, same count) have preceded it without anything in between
    } ra;
    bool inside_run = false;    // ow_run is executing (its own ow_update_all calls are not "something in between")
    int run_frames = 0;         // ... with this many ticks (may_split)
    int pair_dir = 0;           // direction of the next block of the cascade-major pair stream (batches 0 .. B-1 or B-1 .. 0): alternates, across runs too
    bool run_as_calls = false;  // OW_FLAG_RUN_AS_CALLS
    bool run_as_reference = false;  // OW_FLAG_RUN_AS_REFERENCE_SCHEDULE
    bool no_merge = false;      // OW_FLAG_NO_TICK_GROUPS
    int group_depth_forced = 0;  // OW_DEBUG_TICK_GROUP_DEPTH (measurements; read once)
    int run_delta_period = 0;    // OW_DEBUG_RUN_DELTA_CHANGE_EVERY: the call-by-call forms of ow_run (OW_FLAG_RUN_AS_CALLS / _AS_REFERENCE_SCHEDULE) switch
                                 // between delta and 1.25 delta every that many ticks -- an irregular caller for the look-ahead to miss on (measurements)
    int ahead_depth = 0;      // ticks of pass 1 ow_update_all's look-ahead computes per launch once the deltas keep repeating (OW_DEBUG_LOOKAHEAD_DEPTH, read once)
    int pair_tick_block = 0;  // ticks a batch runs through before the stream of tick pairs moves on to the next batch (0: by map size; OW_DEBUG_PAIR_TICK_BLOCK, read once)
    size_t pair_texels = 0;  // batch size of ow_run's tick pairs, in texels (kPairTexels; OW_DEBUG_PAIR_TEXELS is read ONCE, by ow_create)
    // ow_run's tick groups (k_tick_group_c_lp): the largest cascade count they serve (0 = not available) and how many ticks go
    // into one group; the scratch buffers hold 2 * depth * count cascades then
    int group_p1_form = -1, group_p2_form = -1;
    int group_max_count = 0, group_depth = 0;  // (group_depth: the depth of a run of group_max_count cascades; a run's own depth follows its count)
    int scratch_slots = 0;  // launch slots the scratch intermediate (T, pcol, rrow) holds now: one batch at create, grown by the first ow_run that merges launches
    // ow_run's tick pairs on the compact family (k_tick_pair_c): the largest batch they launch (0 = never); scratch two batches deep
    int pair_slots = 0;
    int last_group_depth = 0;  // ticks per launch of the most recent ow_run that went out in groups / pairs
    // timing: a pool of events so that timed ticks stay enqueued back to back
    int timing = 0;  // 0 off, 1 per pass (ow_run stays on one launch per pass), 2 as launched (tick groups / pairs stay on, timed per launch)
    std::vector<char> ev_single;  // per 4-event record: 1 = one launch (events 0, 1 only): a tick group / pair
    std::vector<hipEvent_t> ev;  // 4 per timed batch: start/stop of the pass-1 dispatch, start/stop of the pass-2 dispatch
    size_t ev_used = 0;
    double t1_ms = 0, t2_ms = 0, tg_ms = 0;
    int t_launches = 0, tg_launches = 0;
    int slot_of[OW_MAX_CASCADES];  // launch slot of each cascade in the most recent batch, -1 if it was not in it
    // last batch that was launched (for ow_probe_kernel_times)
    ow::FrameArgs last_args{};
    int last_count = 0;
    // hand-off to a host consumer (ow_readback_*): device snapshot + page-locked staging, one slot per layer and map
    hipStream_t copy_stream = nullptr;
    ow::u16x4 *snap_dev = nullptr, *snap_host = nullptr;  // [2 maps][layers][N][N]
    hipEvent_t snap_ready[OW_MAX_CASCADES] = {}, copy_done[OW_MAX_CASCADES] = {};
    bool copy_pending[OW_MAX_CASCADES] = {};
    // ow_sample_surface scratch (grow-only)
    float *query_xz = nullptr;
    ow::SurfaceSample *query_out = nullptr;
    int query_capacity = 0;
};

// The stream everything but a first-chain launch is enqueued on or synchronised through: joins the second chain first (a no-op when none is in flight).
// A failing event call leaves side_active set and hands back the stream all the same: the enqueue that follows reports the device's state itself.
static hipStream_t main_stream(ow_context *c) {
    if (c->side_active && hipEventRecord(c->side_join_ev, c->side_stream) == hipSuccess && hipStreamWaitEvent(c->stream, c->side_join_ev, 0) == hipSuccess)
        c->side_active = false;
    return c->stream;
}
// before a launch that is split: the second chain starts behind everything enqueued on `stream` so far (its cascades' previous work included)
static bool side_fork(ow_context *c) {
    if (c->side_active) return true;
    if (hipEventRecord(c->side_fork_ev, c->stream) != hipSuccess || hipStreamWaitEvent(c->side_stream, c->side_fork_ev, 0) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    c->side_active = true;
    return true;
}
// every launch of the merged shapes (tick groups, tick pairs): split over the two chains where that pays and the context may
// On a stream of the CALLER's the second chain has to be joined before every call returns (join_for_caller), and a join -- an event record plus a cross-stream wait --
// costs 10 - 20 us: more than a split launch gains (4 us at 1024^2 x 4).  So there only the launches of an ow_run of at least 8 ticks (16 at 512^2) are split (one join per
// run: 1024^2 x 4, K = 10: 51.4 against 52.6 us per tick, K = 5: 53.8 against 53.8; a caller that issues one ow_update_all per tick on its own stream would pay 68 us
// where one stream takes 51, profiles/r06_chain_region_cost.txt); on the context's own stream nothing is joined until something synchronises, and every launch is split.
// (512^2 x 8, ticks half as long: K = 10 29.8 against 29.0, K = 20 27.3 against 27.8 -- sixteen ticks there: ~0.4 ms of work either way.)
static bool may_split(const ow_context *c) {
    const int min_run = c->n >= 1024 ? 8 : 16;
    return c->side_stream && (c->own_stream || (c->inside_run && !c->run_as_calls && !c->run_as_reference && c->run_frames >= min_run));
}
static hipError_t launch_group(ow_context *c, const ow::FrameArgs &args, const ow::TickGroupArgs &ga, const ow::LaunchTiming &lt = ow::LaunchTiming{}) {
    if (may_split(c) && !lt.start && ow::tick_pair_splits(c->n, ga) && side_fork(c)) {
        ++c->split_launches;
        return ow::launch_tick_group(c->n, args, ga, c->buf, c->stream, lt, c->side_stream);
    }
    return ow::launch_tick_group(c->n, args, ga, c->buf, main_stream(c), lt);
}
// a caller that brought its own stream orders its own work behind ours by that stream alone: nothing of the second chain may outlive a call
static void join_for_caller(ow_context *c) {
    if (!c->own_stream) (void)main_stream(c);
}

namespace {

size_t plane(const ow_context *c) { return (size_t)c->n * c->n; }

// Cascades per pair of launches (measured at 1024^2, scripts/mode_bench.py): a tick of up to 6 Mi texels goes in ONE pair
// (x 5: 82 us against 84.5 as 4 + 1; x 6: 88.5 against 96); beyond that the intermediate and the inputs of a pair no
// longer fit the 256 MiB Infinity Cache together (x 8 in one pair: 126-137 us, erratic) and the tick is split into equal
// batches of at most 4 Mi texels (x 7: 4 + 3, x 8: 4 + 4 = 121-125 us; 2048^2: one cascade per pair).
// (2048^2 with 8 or 16 Mi texels per pair, measured in round 2: x 4 310.7 -> 308.1 / 300.9 us per tick, while 1024^2 x 8 gets worse,
// 125.4 -> 136.8: left at 4 Mi.)
constexpr size_t kSinglePairTexels = 6u << 20, kBatchTexels = 4u << 20;
int max_batch(const ow_context *c) {  // the most cascades a pair ever takes: sizes the scratch buffers
    const size_t pl = (size_t)c->n * c->n;
    return std::max(1, (int)std::max(kSinglePairTexels / pl, kBatchTexels / pl));
}
int batch_size(const ow_context *c, int count) {
    const size_t pl = (size_t)c->n * c->n;
    if ((size_t)count * pl <= kSinglePairTexels) return count;
    const int cap = std::max(1, (int)(kBatchTexels / pl)), batches = (count + cap - 1) / cap;
    return (count + batches - 1) / batches;
}

// ow_run on a batch of the layer-parallel compact family goes out in tick groups (k_tick_group_c_lp), which need the scratch
// intermediate 2 * depth times.  Largest cascade count served for this context, and the depth that fits kGroupScratchBytes:
// 512 MiB = four ticks per launch up to 512^2 x 6 (round 4, us per tick at 2 / 3 / 4 / 6 ticks per launch: 512^2 x 5 21.5 / 20.2 / 19.6 / 18.9,
// x 6 25.3 / 24.6 / 23.0 / 23.3: with 256 MiB these ran at three and two; profiles/r04_group_depth.txt).  A few hundred MiB of a 288 GB part.
constexpr size_t kGroupScratchBytes = (size_t)512 << 20;
// Ticks of the compact family go out as tick pairs (k_tick_pair_c / k_tick_pair_c_split: pass 2 of one batch and pass 1 of the next in one
// launch), in equal batches of at most kPairTexels -- a tick of 1024^2 x 5 .. 8 is two batches (3 + 2, 3 + 3, 4 + 3, 4 + 4), at 2048^2 a batch is
// one cascade.  The scratch is two batches deep: at 4 Mi texels 160 MiB of intermediate in flight, which the Infinity Cache holds next to the
// batch's own spectra and foam (run_tick_pairs orders the stream so that a batch shares the cache with nothing but itself).  A single batch
// of 5 or 6 Mi texels loses (profiles/r02_tick_pairs_compact.txt).
constexpr size_t kPairTexels = (size_t)4 << 20;
// the batches of one tick of `count` cascades (sizes[], larger first); 0 = no tick pairs for this count
int pair_batches(const ow_context *c, int count, int *sizes) {
    const size_t pair_texels = c->pair_texels, pl = (size_t)c->n * c->n;
    if (count < 1 || !ow::tick_pairs_supported(c->n)) return 0;
    // Equal batches of at most pair_texels.  (Round 3 halved the batches where two full-size ones next to the spectra overflow the Infinity
    // Cache -- 1024^2 x 8 as four batches of two: 115 -> 113 us per tick in tick-major order.  With the stream in cascade-major order, below,
    // a batch only shares the cache with ITSELF one tick later, and full-size batches win: 108.4 -> 104.9; profiles/r04_pairs_order_1024.txt.)
    const int cap = std::max(1, (int)(pair_texels / pl)), B = (count + cap - 1) / cap;
    for (int b = 0, left = count; b < B; ++b) {
        sizes[b] = (left + (B - b) - 1) / (B - b);
        left -= sizes[b];
        if (ow::kernel_family(c->n, sizes[b], c->kernel_mode) != 3) return 0;
    }
    return B;
}
// ticks per launch of a run of `count` cascades in tick groups: four; eight where a tick is small -- up to 512 Ki texels (256^2 x <= 8,
// 512^2 x <= 2; us per tick at 4 / 8: 256^2 x 4 5.9 / 5.3 before the pipelined pass 2, x 8 7.79 / 7.28, 512^2 x 2 7.85 / 7.35; from 512^2 x 4
// and 1024^2 x 1 on four is the best: 14.3 / 15.0, 14.7 / 15.4) -- never more than fits kGroupScratchBytes twice over (double-buffered)
int tick_group_depth_for(const ow_context *c, int count) {
    const size_t per_tick = (size_t)count * c->n * c->n * ow::kLayers * sizeof(ow::cplx);
    // ... and further where a launch of eight is still mostly fixed costs: sixteen for one or two cascades of up to 256 Ki texels together
    // (us per tick at 8 / 12 / 16: 256^2 x 1 3.10 / 2.85 / 2.79, x 2 3.17 / 3.03 / 2.97, 512^2 x 1 4.21 / 3.99 / 3.89), twelve for three or
    // four (256^2 x 4 3.95 / 3.82 / 3.91); 256^2 x 8 stays at eight (7.41 / 7.21)
    const size_t cap = per_tick <= ((size_t)8 << 20) ? (count <= 2 ? 16 : 12) : per_tick <= ((size_t)16 << 20) ? 8 : 4;
    static_assert(ow::kMaxTickGroup >= 16, "the deepest tick group has to fit TickGroupArgs");
    if (c->group_depth_forced > 0) return c->group_depth_forced;
    return (int)std::min<size_t>(cap, std::max<size_t>(1, kGroupScratchBytes / (2 * per_tick)));
}
void plan_tick_groups(ow_context *c, uint32_t flags) {
    c->group_max_count = c->group_depth = c->pair_slots = 0;
    c->pair_tick_block = 0;
    c->group_depth_forced = 0;
    c->run_delta_period = 0;
    c->ahead_depth = ow_context::Lookahead::kMaxAhead;
    c->pair_texels = kPairTexels;
    // the two forms of the tick groups' work items can be pinned per context (tests hold every form to the same bits at every size)
    c->group_p1_form = (flags & OW_FLAG_GROUP_P1_COMPACT) ? 1 : (flags & OW_FLAG_GROUP_P1_LP) ? 0 : -1;
    c->group_p2_form = (flags & OW_FLAG_GROUP_P2_PIPE) ? 1 : (flags & OW_FLAG_GROUP_P2_PLAIN) ? 0 : -1;
#ifdef OW_MEASUREMENT_KNOBS
    // Measurement knobs of A/B builds (scripts/build_variant.sh knobs -DOW_MEASUREMENT_KNOBS): the SHIPPED library reads nothing from the
    // environment.  Read HERE and nowhere else (ow_create): the scratch is sized from pair_slots, which follows from the batch size, and a value
    // that changed between ow_create and ow_run would let the merged launches write past that scratch.
    //   OW_DEBUG_RUN_DELTA_CHANGE_EVERY  the call-by-call forms of ow_run switch between delta and 1.25 delta every that many ticks (changes RESULTS)
    //   OW_DEBUG_TICK_GROUP_DEPTH        ticks per launch of the tick groups        OW_DEBUG_LOOKAHEAD_DEPTH  ticks of pass 1 computed ahead at most
    //   OW_DEBUG_PAIR_TICK_BLOCK         ticks a batch runs through before the stream of tick pairs moves on (1 = tick-major)
    //   OW_DEBUG_PAIR_TEXELS             batch size of the tick pairs in Mi texels
    if (const char *e = getenv("OW_DEBUG_RUN_DELTA_CHANGE_EVERY")) c->run_delta_period = std::max(0, std::min(1 << 20, atoi(e)));
    if (const char *e = getenv("OW_DEBUG_TICK_GROUP_DEPTH")) c->group_depth_forced = std::max(0, std::min((int)ow::kMaxTickGroup, atoi(e)));
    if (const char *e = getenv("OW_DEBUG_LOOKAHEAD_DEPTH")) c->ahead_depth = std::max(1, std::min((int)ow_context::Lookahead::kMaxAhead, atoi(e)));
    if (const char *e = getenv("OW_DEBUG_PAIR_TICK_BLOCK")) c->pair_tick_block = std::max(0, std::min(4096, atoi(e)));
    if (const char *e = getenv("OW_DEBUG_PAIR_TEXELS"))
        if (atol(e) >= 1 && atol(e) <= 64) c->pair_texels = (size_t)atol(e) << 20;
#endif
    if ((flags & OW_FLAG_NO_TICK_GROUPS) || !ow::tick_pairs_supported(c->n)) return;
    for (int count = 1; count <= c->cascades; ++count) {
        int sizes[OW_MAX_CASCADES];
        if (pair_batches(c, count, sizes) > 0) c->pair_slots = std::max(c->pair_slots, sizes[0]);
    }
    int best = 0;
    for (int count = 1; count <= c->cascades; ++count)
        if (ow::kernel_family(c->n, count, c->kernel_mode) == 4) best = count;
    if (best == 0) return;
    c->group_max_count = best;
    c->group_depth = tick_group_depth_for(c, best);
}
// One batch of scratch intermediate is what a tick launched one pass at a time needs.  The look-ahead of ow_update_all / ow_process keeps
// more in flight -- the pair kernel two batches, the group kernel a ring of five groups -- and ow_create allocates THAT (lookahead_scratch_slots):
// the per-frame calls never allocate, so memory use is what ow_create left behind and no frame pays a hipMalloc + synchronize (round 4 grew the
// scratch inside the first speculating call).  Only ow_run -- the throughput form, whose merged launches keep 2 * depth ticks (groups: up to
// kGroupScratchBytes) in flight -- still grows it on first use.
int base_scratch_slots(const ow_context *c) { return std::min(c->layers, max_batch(c)); }
ow_status ensure_scratch(ow_context *c, int slots) {
    if (slots <= c->scratch_slots) return OW_OK;
    const size_t pl = (size_t)c->n * c->n;
    ow::cplx *T = nullptr, *pcol = nullptr, *rrow = nullptr;
    if (hipMalloc((void **)&T, (size_t)slots * pl * ow::kLayers * sizeof(ow::cplx)) != hipSuccess ||
        hipMalloc((void **)&pcol, (size_t)slots * c->n * sizeof(ow::cplx)) != hipSuccess ||
        hipMalloc((void **)&rrow, (size_t)slots * c->n * 4 * sizeof(ow::cplx)) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(T);
        (void)hipFree(pcol);
        (void)hipFree(rrow);
        return fail(OW_ERR_NOMEM, "hipMalloc failed for %d launch slots of scratch intermediate (%zu bytes)", slots, (size_t)slots * pl * ow::kLayers * sizeof(ow::cplx));
    }
    // the old scratch may still be in use by launches already enqueued (it is dead once they have finished: scratch of one batch)
    if (c->buf.T && hipStreamSynchronize(main_stream(c)) != hipSuccess) {
        (void)hipFree(T);
        (void)hipFree(pcol);
        (void)hipFree(rrow);
        return fail(OW_ERR_HIP, "hipStreamSynchronize failed while growing the scratch intermediate");
    }
    (void)hipFree(c->buf.T);
    (void)hipFree(c->buf.pcol);
    (void)hipFree(c->buf.rrow);
    c->buf.T = T;
    c->buf.pcol = pcol;
    c->buf.rrow = rrow;
    c->scratch_slots = slots;
    c->la.armed = false;                 // (a speculated pass 1 went with the old buffer, too)
    c->ra.armed = false;
    for (int &sl : c->slot_of) sl = -1;  // (the reference-layout view of the last batch's intermediate went with the old buffer)
    return OW_OK;
}

// The runtime carves allocations below 2 MiB out of shared 2 MiB buffer objects, and a dma-buf always covers the whole object
// (measured: scripts/dmabuf_probe.py -- an import of the second 1 MiB allocation saw the first one's bytes).  The two output arrays
// are therefore allocated in whole multiples of 2 MiB: each is a buffer object of its own and ow_export_maps' descriptors map it
// from offset 0.
constexpr size_t kExportGranule = (size_t)2 << 20;
size_t exportable_bytes(size_t bytes) { return (bytes + kExportGranule - 1) / kExportGranule * kExportGranule; }

constexpr size_t kMaxTimedBatches = 4096;

ow_status collect_timing(ow_context *c) {
    if (c->ev_used == 0) return OW_OK;
    OW_HIP(hipEventSynchronize(c->ev[c->ev_used - 1]));
    for (size_t i = 0; i + 4 <= c->ev_used; i += 4) {
        float a = 0, b = 0;
        OW_HIP(hipEventElapsedTime(&a, c->ev[i], c->ev[i + 1]));
        if (i / 4 < c->ev_single.size() && c->ev_single[i / 4]) {
            c->tg_ms += a;
            c->tg_launches += 1;
            continue;
        }
        OW_HIP(hipEventElapsedTime(&b, c->ev[i + 2], c->ev[i + 3]));
        c->t1_ms += a;
        c->t2_ms += b;
        c->t_launches += 1;
    }
    c->ev_used = 0;
    return OW_OK;
}

ow_status next_events(ow_context *c, hipEvent_t **out, bool single = false) {
    if (c->ev_used + 4 > 4 * kMaxTimedBatches) {
        ow_status st = collect_timing(c);
        if (st != OW_OK) return st;
    }
    while (c->ev.size() < c->ev_used + 4) {
        hipEvent_t e;
        OW_HIP(hipEventCreate(&e));
        c->ev.push_back(e);
    }
    *out = &c->ev[c->ev_used];
    if (c->ev_single.size() < c->ev_used / 4 + 1) c->ev_single.resize(c->ev_used / 4 + 1);
    c->ev_single[c->ev_used / 4] = single ? 1 : 0;
    c->ev_used += 4;
    return OW_OK;
}

// the status word, consumed: marks what the faulted batches left behind (shared by sync_stream and ow::poll_status)
ow_status consume_status(ow_context *c) {
    if (!c->status_host || *c->status_host == 0u) {
        return OW_OK;
    }
    const uint32_t bits = *c->status_host;
    *c->status_host = 0u;                          // the word is consumed; later batches start clean ...
    c->maps_faulted |= c->enqueued_since_sync;     // ... but what the faulted batches left behind stays marked until those layers are recomputed
    c->enqueued_since_sync = 0;
    // ... and so does whatever those launches computed AHEAD: a speculated pass 1 (of cascades that are not even enqueued yet, or of the same
    // cascade one tick later at 2048^2, where pass 1 has a bounded wait of its own) may be as corrupt as the maps.  The queue is dropped: the
    // ticks after a fault recompute their pass 1.
    c->la.armed = false;
    c->la.queued = 0;
    c->ra.armed = false;
    for (int i = 0; i < OW_MAX_CASCADES; ++i)
        if (c->copy_pending[i]) c->readback_faulted |= 1u << i;
    return fail(OW_ERR_HIP, "device-side failure reported by a frame kernel (status 0x%x%s): the maps of the batches enqueued since "
                            "the last synchronisation are invalid", bits, (bits & ow::kStatusRowSyncTimeout) ? ": wave-pair rendezvous timed out" : "");
}
ow_status refuse_faulted(const ow_context *c, uint32_t layer_mask) {
    if (c->maps_faulted & layer_mask)
        return fail(OW_ERR_HIP, "layer mask 0x%x holds maps of a batch that reported a device-side failure (already returned by an earlier call); "
                                "they stay invalid until a later batch has recomputed those layers (faulted: 0x%x)", layer_mask, c->maps_faulted);
    return OW_OK;
}
// hipStreamSynchronize + the device status word.  layer_mask: the array layers whose bytes the caller is about to hand to ITS caller (0 for a
// bare ow_sync), refused while they are those of a faulted batch.
ow_status sync_stream(ow_context *c, uint32_t layer_mask) {
    OW_HIP(hipStreamSynchronize(main_stream(c)));
    if (ow_status st = consume_status(c); st != OW_OK) return st;
    c->enqueued_since_sync = 0;  // everything enqueued so far has finished cleanly
    return refuse_faulted(c, layer_mask);
}

}  // namespace
namespace ow {
// The device status word WITHOUT synchronising: for a caller that has waited by other means (the group's gather waits on its own
// copy events) and is about to hand out map bytes.  Same consumption / stickiness as sync_stream.
ow_status poll_status(ow_context *c) {
    if (ow_status st = consume_status(c); st != OW_OK) return st;
    return refuse_faulted(c, (1u << c->cascades) - 1u);
}
}  // namespace ow
namespace {

// every field finite -- also AFTER the narrowing of the push-constant pack (an FP64 value beyond FP32's range would reach the kernels as inf)
bool finite_record(const ow_cascade_params &p) {
    const double f[] = {p.tile_length[0], p.tile_length[1], p.wind_speed, p.wind_direction, p.fetch_length, p.swell, p.spread, p.detail,
                        p.whitecap, p.foam_amount, p.displacement_scale, p.normal_scale};
    for (double v : f)
        if (!std::isfinite(v) || !std::isfinite((float)v)) return false;
    return std::isfinite(p.time) && std::isfinite(p.foam_grow_rate) && std::isfinite(p.foam_decay_rate);
}

// A record the kernels can take: every field finite, tile_length positive.  Checked where a record ENTERS the context (ow_update,
// ow_set_cascade_params) -- before anything of the caller's is changed and before anything is armed -- so that a refused record
// leaves no trace; enqueue() repeats the check as a guard only.
ow_status validate_record(const ow_cascade_params &p, int index) {
    if (!finite_record(p)) return fail(OW_ERR_INVALID, "cascade %d: non-finite parameter", index);
    if (!(p.tile_length[0] > 0.0f) || !(p.tile_length[1] > 0.0f)) return fail(OW_ERR_INVALID, "cascade %d: tile_length must be positive", index);
    return OW_OK;
}
}  // namespace
namespace ow {
// what ow_update checks before it changes anything (a group checks ALL shards' records with it before any shard starts)
ow_status validate_records(const ow_cascade_params *params, int count, double delta) {
    if (!std::isfinite(delta)) return fail(OW_ERR_INVALID, "delta is not finite");
    for (int i = 0; i < count; ++i) {
        if (ow_status st = validate_record(params[i], i); st != OW_OK) return st;
        if (!std::isfinite(params[i].time + delta)) return fail(OW_ERR_INVALID, "cascade %d: time + delta is not finite", i);
    }
    return OW_OK;
}
}  // namespace ow
namespace {

uint32_t f32_word(float v) {
    uint32_t w;
    std::memcpy(&w, &v, 4);
    return w;
}
// The spectrum block of one cascade (wave_generator.gd:69-71 through render_context.gd:122-135): the launch constants and the thirteen words the
// reference would have packed, in its order.  The exported setters clamp wind_speed and fetch_length (wave_cascade_parameters.gd:15,20:
// max(0.0001, value), FP64); a C caller has no setter.  Everything up to the pack is FP64, as in GDScript (wave_generator.gd:69-71); the
// narrowing is the pack's (render_context.gd:134).
void pack_spectrum(const ow_context *c, int cascade, const ow_cascade_params &p, ow::SpectrumPC &pc, uint32_t (&w)[16]) {
    const double wind_speed = std::max(p.wind_speed, 1e-4), fetch_length = std::max(p.fetch_length, 1e-4);
    const double F = fetch_length * 1e3;
    pc.seed_x = p.spectrum_seed[0];
    pc.seed_y = p.spectrum_seed[1];
    pc.tile_x = p.tile_length[0];
    pc.tile_y = p.tile_length[1];
    pc.alpha = (float)ow_jonswap_alpha(wind_speed, F);
    pc.peak_frequency = (float)ow_jonswap_peak_angular_frequency(wind_speed, F);
    pc.wind_speed = (float)wind_speed;
    pc.angle = (float)(p.wind_direction * (3.14159265358979323846 / 180.0));  // deg_to_rad
    pc.depth = c->depth;
    pc.swell = (float)p.swell;
    pc.detail = (float)p.detail;
    pc.spread = (float)p.spread;
    std::memset(w, 0, sizeof(w));
    w[0] = (uint32_t)pc.seed_x, w[1] = (uint32_t)pc.seed_y, w[2] = f32_word(pc.tile_x), w[3] = f32_word(pc.tile_y);
    w[4] = f32_word(pc.alpha), w[5] = f32_word(pc.peak_frequency), w[6] = f32_word(pc.wind_speed), w[7] = f32_word(pc.angle);
    w[8] = f32_word(pc.depth), w[9] = f32_word(pc.swell), w[10] = f32_word(pc.detail), w[11] = f32_word(pc.spread), w[12] = (uint32_t)cascade;
}
// A DIRTY RECORD WHOSE SPECTRUM IS ALREADY THERE.  In the reference EVERY exported setter raises should_generate_spectrum -- whitecap and
// foam_amount included (wave_cascade_parameters.gd:32-35), which spectrum_compute.glsl never sees -- and _update re-dispatches
// spectrum_compute with the same push constants (wave_generator.gd:68-72): a slider drag costs a spectrum per cascade per update.  The
// spectrum is a deterministic function of the thirteen packed words (and the map size), so when a dirty record packs to exactly the words
// layer `cascade`'s resident spectrum was generated from, regenerating would write the same bits: the flag is simply consumed.  That also
// keeps such a record on the merged launches and the look-ahead, which step aside for a spectrum that has to be generated.
bool spectrum_is_resident(const ow_context *c, int cascade, const ow_cascade_params &p) {
    if (c->always_regenerate || cascade < 0 || cascade >= OW_MAX_CASCADES || !c->spectrum_resident[cascade]) return false;
    ow::SpectrumPC pc;
    uint32_t w[16];
    pack_spectrum(c, cascade, p, pc, w);
    return std::memcmp(w, c->pc_words[cascade].spectrum, sizeof(w)) == 0;
}
// consumes the dirty flag of armed record `cascade` if its spectrum is resident (where a record ENTERS the context: ow_update, ow_set_cascade_params)
void settle_dirty_flag(ow_context *c, int cascade, ow_cascade_params &p) {
    if (p.should_generate_spectrum && finite_record(p) && spectrum_is_resident(c, cascade, p)) {
        p.should_generate_spectrum = 0;
        ++c->spectra_skipped;
    }
}

// render_context.gd:122-135 for the modulate and unpack blocks of one cascade (wave_generator.gd:73,85)
void record_frame_constants(ow_context *c, int cascade, const ow_cascade_params &p) {
    ow_push_constants &w = c->pc_words[cascade];
    std::memset(w.modulate, 0, sizeof(w.modulate));
    std::memset(w.unpack, 0, sizeof(w.unpack));
    w.modulate[0] = f32_word(p.tile_length[0]);
    w.modulate[1] = f32_word(p.tile_length[1]);
    w.modulate[2] = f32_word(c->depth);
    w.modulate[3] = f32_word((float)p.time);
    w.modulate[4] = (uint32_t)cascade;
    w.unpack[0] = (uint32_t)cascade;
    w.unpack[1] = f32_word((float)p.whitecap);
    w.unpack[2] = f32_word((float)p.foam_grow_rate);
    w.unpack[3] = f32_word((float)p.foam_decay_rate);
    c->pc_valid[cascade] = true;
}

// _update() for a batch of cascade indices (wave_generator.gd:65-85)
ow::CascadeFrame frame_of(const ow_cascade_params &p, int cascade) {
    ow::CascadeFrame cf;
    std::memset(&cf, 0, sizeof(cf));
    cf.tile_x = p.tile_length[0];
    cf.tile_y = p.tile_length[1];
    cf.time = (float)p.time;  // push constants are FP32 (render_context.gd:131-134)
    cf.whitecap = (float)p.whitecap;
    cf.foam_grow_rate = (float)p.foam_grow_rate;
    cf.foam_decay = expf(-(float)p.foam_decay_rate);  // fft_unpack.glsl:62, uniform over the dispatch
    cf.cascade = cascade;
    return cf;
}

ow_status enqueue(ow_context *c, ow_cascade_params *params, const int *idx, int count) {
    if (count <= 0) return OW_OK;
    c->la.armed = false;  // this path uses the scratch intermediate from slot 0 on: a speculated pass 1 (lookahead_tick) does not survive it
    c->ra.armed = false;  // ... nor does what a run computed ahead for the next run
    ow::FrameArgs args;
    std::memset(&args, 0, sizeof(args));
    // everything is validated before anything is launched: a bad record must not leave the batch half enqueued
    for (int i = 0; i < count; ++i)
        if (ow_status st = validate_record(params[idx[i]], idx[i]); st != OW_OK) return st;
    for (int i = 0; i < count; ++i) {
        // a new batch recomputes THESE layers' maps (the foam state a faulted batch left behind is the caller's to restore); the other
        // layers keep what they hold, marks included
        c->maps_faulted &= ~(1u << idx[i]);
        c->enqueued_since_sync |= 1u << idx[i];
        ow_cascade_params &p = params[idx[i]];
        if (p.should_generate_spectrum) {  // :68-72
            ow::SpectrumPC pc;
            uint32_t w[16];
            pack_spectrum(c, idx[i], p, pc, w);
            if (!c->always_regenerate && c->spectrum_resident[idx[i]] && std::memcmp(w, c->pc_words[idx[i]].spectrum, sizeof(w)) == 0) {
                ++c->spectra_skipped;  // the same thirteen words: the resident spectrum IS what the dispatch would write (spectrum_is_resident)
            } else {
                c->spectrum_resident[idx[i]] = false;
                OW_HIP(ow::launch_spectrum(c->n, idx[i], pc, c->buf, main_stream(c)));
                std::memcpy(c->pc_words[idx[i]].spectrum, w, sizeof(w));  // wave_generator.gd:71, in the reference's order
                c->spectrum_resident[idx[i]] = true;
                ++c->spectra_generated;
            }
            p.should_generate_spectrum = 0;
        }
        record_frame_constants(c, idx[i], p);
        args.c[i] = frame_of(p, idx[i]);
    }
    // Launch in batches of batch_size() cascades (see there).  Cascades are independent, so batching does not change any result.
    const int per_batch = batch_size(c, count);
    for (int b0 = 0; b0 < count; b0 += per_batch) {
        const int nb = std::min(per_batch, count - b0);
        ow::FrameArgs part;
        std::memset(&part, 0, sizeof(part));
        for (int i = 0; i < nb; ++i) part.c[i] = args.c[b0 + i];
        part.c[0].fault = (int32_t)c->inject_fault;  // debug hook (ow_debug_inject_fault): this batch only
        c->inject_fault = 0;
        c->last_args = part;
        c->last_args.c[0].fault = 0;
        c->last_count = nb;
        c->last_family = ow::kernel_family(c->n, nb, c->kernel_mode);
        for (int &sl : c->slot_of) sl = -1;
        for (int i = 0; i < nb; ++i) c->slot_of[part.c[i].cascade] = i;
        hipEvent_t *ev = nullptr;
        if (c->timing) {
            ow_status st = next_events(c, &ev);
            if (st != OW_OK) return st;
        }
        const ow::LaunchTiming t1{ev ? ev[0] : nullptr, ev ? ev[1] : nullptr}, t2{ev ? ev[2] : nullptr, ev ? ev[3] : nullptr};
        OW_HIP(ow::launch_pass1(c->n, nb, c->kernel_mode, part, c->buf, main_stream(c), t1));  // modulate + rows + transpose (:73-80)
        OW_HIP(ow::launch_pass2(c->n, nb, c->kernel_mode, part, c->buf, main_stream(c), t2));  // rows + unpack (:82-85)
    }
    return OW_OK;
}

// ---- ow_update_all's adaptive look-ahead ------------------------------------------------------------------------------------------
// ow_run knows the ticks to come and merges launches across them (tick groups, tick pairs).  A caller that issues ow_update_all tick by tick
// -- the reference's schedule in throughput form -- gives the runtime no such knowledge, but a regular one is predictable: pass 1 of a tick
// depends on nothing but the spectra, the tile lengths and the FP32 time, and the time of the next tick is this tick's plus the next delta.
// So once TWO consecutive calls have come with the same delta, ow_update_all launches pass 2 of its tick together with a SPECULATED pass 1 of
// the next one (time + delta; the pair / group kernels of ow_run with one tick per side) into the other half of the scratch intermediate.
// The next call checks the speculation against what it was actually given -- cascade count, every slot's FP32 time and tile lengths, bit for
// bit; no spectrum to regenerate; nothing else has used the scratch since -- and on a hit its pass 1 is already there: the tick costs one
// merged launch instead of two (1024^2 x 4: 58.0 -> 54.3 us, 1024^2 x 2: 39.0 -> 28.4).  On a miss the speculated work is discarded and the
// tick takes the ordinary two launches; a caller whose deltas jitter (water.gd's rate limiter passes the elapsed time) never arms it.  Results
// are bit-identical either way (same item bodies; tests/test_lookahead.py).  Single-batch ticks only: a second batch would need its own
// two intermediates.  Off under OW_FLAG_NO_TICK_GROUPS, per-launch timing and fault injection.
// FOUR AHEAD: where the launch is the group kernel (layer-parallel compact family, ticks of up to 1 Mi texels), one launch computes pass 1 of
// as many of the next ticks as the caller's cadence predicts (predicted_repeats), up to four -- a queue of entries, each with its own group of the scratch ring (five groups:
// the one being read + four) -- and the three calls in between launch pass 2 alone: 256^2 x 4 15.3 -> 9.7 us per tick (11.1 with one tick
// ahead), 512^2 x 4 26.8 -> 20.2 (26.2), 1024^2 x 1 29.9 -> 20.3 (25.8); nothing more beyond four (profiles/r04_lookahead_depth.txt).
// The reference's own schedule gets the same without any guessing: ow_process of cascade i KNOWS the cascades the next ow_process calls will
// take (i - 1, i - 2, .., armed with their records), so its launch carries pass 2 of i and pass 1 of up to four of them (the queue's entries are
// then different cascades: TickGroupArgs.step1); only the steps into the next update are speculation (time + the last delta, once the deltas
// repeat).  1024^2 x 4 on that schedule: 119.0 -> 85.3 us per update.
int lookahead_mode(const ow_context *c, int count) {
    if (c->no_merge || c->timing || c->inject_fault) return 0;
    const int fam = ow::kernel_family(c->n, count, c->kernel_mode);
    int sizes[OW_MAX_CASCADES];
    if (fam == 3 && c->pair_slots > 0 && pair_batches(c, count, sizes) == 1) return 1;
    if (fam == 4 && ow::tick_groups_supported(c->n) && count <= c->group_max_count) return 2;
    return 0;
}
// launch slots of scratch the look-ahead can ever ask for in this context (lookahead_launch: groups * stride), over every cascade count a
// call may come with -- ow_update_all with 1 .. cascades, ow_process with one
int lookahead_scratch_slots(const ow_context *c) {
    int slots = 0;
    for (int count = 1; count <= c->cascades; ++count) {
        const int mode = lookahead_mode(c, count);
        if (mode == 1) slots = std::max(slots, 2 * c->pair_slots);
        if (mode == 2) slots = std::max(slots, (ow_context::Lookahead::kMaxAhead + 1) * count);
    }
    return slots;
}
// What a caller wants launched: pass 2 of `now_count` cascades (launch-slot order) and, ahead of time, pass 1 of the cascades its next
// `ahead_ticks` launches will take at the FP32 times they WILL be processed with -- KNOWN where the cascades are armed (ow_process: the next
// cascades of the same update), SPECULATED where they belong to a tick the caller has not issued yet (time + the caller's last delta).
struct LookaheadPlan {
    int now[OW_MAX_CASCADES], now_count;
    int next_count;                          // 0, or now_count: cascades per launch computed ahead
    int ahead_ticks;                         // how many of the caller's next launches may be computed ahead
    int next[ow_context::Lookahead::kMaxAhead][OW_MAX_CASCADES];  // their cascades: the same ones a tick later each (ow_update_all), or the ones after `now` (ow_process)
    float next_time[ow_context::Lookahead::kMaxAhead][OW_MAX_CASCADES];
};
// returns true if the launch has been made here (status in *out); false: the caller takes the ordinary path
bool lookahead_launch(ow_context *c, const LookaheadPlan &pl, ow_status *out) {
    ow_context::Lookahead &la = c->la;
    constexpr int kRing = ow_context::Lookahead::kMaxAhead;
    const int count = pl.now_count, mode = lookahead_mode(c, count);
    bool eligible = mode != 0;
    for (int i = 0; eligible && i < count; ++i) {
        const ow_cascade_params &p = c->pass_parameters[pl.now[i]];
        eligible = !p.should_generate_spectrum && validate_record(p, pl.now[i]) == OW_OK;
    }
    bool hit = la.armed && la.queued > 0 && eligible && la.count == count && la.mode == mode;
    for (int i = 0; hit && i < count; ++i) {
        const ow_cascade_params &p = c->pass_parameters[pl.now[i]];
        const float t = (float)p.time;
        hit = la.cascade[la.head][i] == pl.now[i] && std::memcmp(&t, &la.time[la.head][i], 4) == 0 && p.tile_length[0] == la.tile_x[la.head][i] && p.tile_length[1] == la.tile_y[la.head][i];
    }
    // the launch slots of the cascades computed ahead: the very slots of `now` where they are the same cascades (the next ticks of an
    // ow_update_all), slots of their own behind them otherwise (the cascades the next ow_process calls will take)
    bool ahead = eligible && pl.next_count == count && pl.ahead_ticks >= 1 && !la.hold;
    bool same = ahead;
    for (int k = 0; ahead && k < std::min(pl.ahead_ticks, kRing); ++k)
        for (int i = 0; i < count; ++i) same = same && pl.next[k][i] == pl.now[i];
    // 2048^2: pass 1 of ANOTHER cascade beside pass 2 is the tick-major pairing that loses there (two cascades' spectra and intermediates do not
    // share the Infinity Cache: 126.0 -> 138.1 us per tick of two cascades on the reference's schedule, profiles/r04_lookahead.txt); the same
    // cascade one tick later is fine (62.5 -> 59.8)
    if (ahead && !same && c->n >= 2048) ahead = false;
    // how many launches ahead: the group kernel takes several ticks of pass 1 per launch (the following calls then launch pass 2 alone), the pair
    // kernel one; never beyond what is asked for, what validates, or what fits the launch's eight cascade frames
    int depth = !ahead ? 0 : std::min(pl.ahead_ticks, mode == 2 ? c->ahead_depth : 1);
    if (!same) depth = std::min(depth, (OW_MAX_CASCADES - count) / count);
    for (int k = 0; k < depth; ++k)
        for (int i = 0; i < count; ++i) {
            const ow_cascade_params &p = c->pass_parameters[pl.next[k][i]];
            if (p.should_generate_spectrum || validate_record(p, pl.next[k][i]) != OW_OK || !std::isfinite(pl.next_time[k][i])) depth = k;
        }
    ahead = depth >= 1;
    const int first1 = (ahead && !same) ? count : 0, step1 = first1;
    const int stride = mode == 1 ? c->pair_slots : count, groups = (mode == 2 ? kRing : 1) + 1;
    if (ahead && ensure_scratch(c, groups * stride) != OW_OK) ahead = false;  // (growing the scratch disarms: checked before `hit` is used)
    hit = hit && la.armed;
    if (!hit) la.queued = 0;  // whatever was computed ahead was computed for something else
    const bool refill = ahead && (hit ? la.queued == 1 : true);  // (a hit that leaves entries queued launches its pass 2 alone)
    if (!hit && !refill) {
        la.armed = false;
        return false;
    }
    c->ra.armed = false;  // (the launches below write the scratch: whatever a run had computed ahead for a next run is gone)
    ow::FrameArgs args;
    std::memset(&args, 0, sizeof(args));
    for (int i = 0; i < count; ++i) {
        const int cascade = pl.now[i];
        const ow_cascade_params &p = c->pass_parameters[cascade];
        c->maps_faulted &= ~(1u << cascade);
        c->enqueued_since_sync |= 1u << cascade;
        record_frame_constants(c, cascade, p);
        args.c[i] = frame_of(p, cascade);
    }
    c->last_args = args;  // (the probe relaunches THIS tick's cascades)
    c->last_count = count;
    c->last_family = ow::kernel_family(c->n, count, c->kernel_mode);
    for (int &sl : c->slot_of) sl = -1;
    auto launched = [&](hipError_t e) {
        if (e == hipSuccess) return true;
        la.armed = false;
        la.queued = 0;
        *out = fail(OW_ERR_HIP, "look-ahead launch failed: %s", hipGetErrorString(e));
        return false;
    };
    int cur = 0;  // scratch group of this launch's own intermediate
    if (!hit) {   // it has to be computed now: the ordinary launch, scratch slots 0 .. = group 0
        if (!launched(ow::launch_pass1(c->n, count, c->kernel_mode, args, c->buf, main_stream(c)))) return true;
    } else {
        ++la.hits;
        cur = la.group[la.head];
        la.head = (la.head + 1) % kRing;
        --la.queued;
    }
    ow::TickGroupArgs ga;
    std::memset(&ga, 0, sizeof(ga));
    if (refill) {  // (the queue is empty here: every group but `cur` is free)
        la.head = 0;
        for (int k = 0; k < depth; ++k) {
            la.group[k] = (cur + 1 + k) % groups;
            ga.tbase1[k] = la.group[k] * stride;
            for (int i = 0; i < count; ++i) {
                const ow_cascade_params &p = c->pass_parameters[pl.next[k][i]];
                if (!same) args.c[first1 + k * step1 + i] = frame_of(p, pl.next[k][i]);  // (pass 1 takes the tile lengths and the layer from it; the times from time1)
                la.time[k][i] = ga.time1[k][first1 + i] = pl.next_time[k][i];
                la.cascade[k][i] = pl.next[k][i];
                la.tile_x[k][i] = p.tile_length[0];
                la.tile_y[k][i] = p.tile_length[1];
            }
        }
        la.queued = depth;
    }
    ga.tbase2[0] = cur * stride;
    ga.d2 = 1;
    ga.d1 = refill ? depth : 0;
    ga.first1 = first1;
    ga.step1 = mode == 2 ? step1 : 0;
    if (mode == 1) {
        ga.pair_compact = 1;
        ga.slots2 = count;
        ga.slots1 = refill ? count : 0;
    } else {
        ga.slots = count;
        // pass-1 items in the layer-parallel form: with ONE tick of pass 2 per launch the launch is as empty as a lone tick, where more and smaller
        // blocks win (measured, us per tick, one launch per pass | look-ahead with lp items | with k_pass1c-shaped items: 256^2 x 4 15.3 | 11.1 |
        // 15.9; 512^2 x 1 15.6 | 11.4 | 17.0; 512^2 x 4 27.3 | 26.2 | 28.4; 1024^2 x 1 29.9 | 25.8 | 28.1; profiles/r04_lookahead.txt)
        // ... and in k_pass1c's form from 512 Ki texels per tick on, where the several ticks of pass 1 of one launch are enough work for the fewer,
        // larger items (us per tick at four ticks ahead, lp | compact: 256^2 x 4 9.7 | 10.5, 512^2 x 1 10.2 | 11.2, 256^2 x 8 13.3 | 12.2,
        // 512^2 x 2 13.7 | 12.7, 512^2 x 4 23.9 | 19.7, 1024^2 x 1 23.0 | 19.7; profiles/r04_lookahead_depth.txt)
        ga.p1_compact = c->group_p1_form >= 0 ? c->group_p1_form : ((size_t)count * depth * c->n * c->n >= ((size_t)2 << 20) ? 1 : 0);
    }
    if (!launched(launch_group(c, args, ga))) return true;
    la.armed = la.queued > 0;
    la.count = count;
    la.mode = mode;
    la.cur_group = cur;
    la.speculated += refill ? 1 : 0;
    *out = OW_OK;
    return true;
}
// How many further updates with the same delta the caller's cadence lets one assume.  The delta has repeated `streak` times so far; the
// caller's PREVIOUS run of equal deltas was prev_run updates long.  Inside a run that the previous one predicts, no further than that one
// went (a caller whose delta changes every k updates is never speculated across a change: measured before this rule, such callers paid up
// to 30 % MORE than one launch per pass -- every change threw away up to four ticks of pass 1; profiles/r04_lookahead_misses.txt); beyond
// it -- and for the first run, prev_run = 1 -- as many as this run has outlasted the prediction by: evidence accumulates anew.
int predicted_repeats(const ow_context::Lookahead &la) {
    const int s = la.streak, last = la.prev_run - 1;  // `last`: the streak value at which the previous run ended
    if (s < 1) return 0;
    return s < last ? std::min(s, last - s) : s - last;
}
// ow_update_all: this tick's cascades now, the same cascades ahead -- one tick once the caller's deltas repeat, several once they keep repeating
bool lookahead_tick(ow_context *c, double delta, int count, ow_status *out) {
    LookaheadPlan pl;
    std::memset(&pl, 0, sizeof(pl));
    pl.now_count = count;
    const int repeats = std::isfinite(delta) ? std::max(predicted_repeats(c->la), c->la.certain) : 0;
    const bool speculate = repeats >= 1;
    pl.next_count = speculate ? count : 0;
    // as many ticks ahead as the caller's cadence predicts (up to four)
    pl.ahead_ticks = !speculate ? 0 : std::min(repeats, (int)ow_context::Lookahead::kMaxAhead);
    for (int i = 0; i < count; ++i) {  // launch slot i = cascade count - 1 - i, as enqueue() takes them from ow_update_all
        pl.now[i] = count - 1 - i;
        double t = c->pass_parameters[count - 1 - i].time;
        for (int k = 0; k < ow_context::Lookahead::kMaxAhead; ++k) {
            t += delta;  // what the k + 1-th ow_update from now will make of it (wave_generator.gd:103: one FP64 add per update), narrowed by the pack
            pl.next[k][i] = count - 1 - i;
            pl.next_time[k][i] = (float)t;
        }
    }
    return lookahead_launch(c, pl, out);
}
// ow_process of armed cascade `idx`: ahead go the cascades the NEXT ow_process calls will take -- idx - 1, idx - 2, .. with their armed records
// (known, not guessed), or, behind the update's last cascade, the cascades of the next update at time + delta once the caller's deltas repeat
bool lookahead_process(ow_context *c, int idx, ow_status *out) {
    constexpr int kMax = ow_context::Lookahead::kMaxAhead;
    LookaheadPlan pl;
    std::memset(&pl, 0, sizeof(pl));
    pl.now_count = 1;
    pl.now[0] = idx;
    // the caller's next launches, in order: the armed cascades below idx (known), then -- a guess, once the deltas repeat -- the cascades of the
    // next update(s) from the top, each at its time + delta (+ delta ...)
    int k = 0;
    for (; k < kMax && k < idx; ++k) {
        pl.next[k][0] = idx - 1 - k;
        pl.next_time[k][0] = (float)c->pass_parameters[idx - 1 - k].time;
    }
    const int pc = c->pass_count;
    const int repeats = predicted_repeats(c->la);
    for (int j = 0; k < kMax && pc >= 1 && j / pc < repeats; ++j, ++k) {  // (update j / pc + 1 from now: only as far as the cadence predicts)
        const int cascade = pc - 1 - j % pc;
        double t = c->pass_parameters[cascade].time;
        for (int r = 0; r <= j / pc; ++r) t += c->la.last_delta;  // (one FP64 add per update, wave_generator.gd:103)
        pl.next[k][0] = cascade;
        pl.next_time[k][0] = (float)t;
    }
    pl.next_count = k > 0 ? 1 : 0;
    pl.ahead_ticks = k;
    return lookahead_launch(c, pl, out);
}

// ow_update on the reference's schedule -- ow_process calls will follow, one cascade each, highest index first: pass 1 of the cascades those calls
// will take is launched NOW, all in ONE launch (the group kernel's pass-1 items, one cascade per "tick": up to four cascades = up to 4 Mi texels,
// a launch that fills the chip), and every ow_process then launches its pass 2 alone.  Nothing is guessed: the records are the armed ones, and each
// ow_process checks what it finds (a live edit in between simply takes the ordinary two launches).  This is what a caller whose deltas never repeat
// gets -- water.gd's rate limiter passes the elapsed time -- where the look-ahead ACROSS updates cannot arm: before (round 4) the first ow_process
// of an update launched pass 1 of its own cascade alone (a quarter-filled launch), then pass 2 + the others' pass 1; and whatever an update left
// for the next one to flush was recomputed from scratch although its pass 1 had long been done (the flush now consumes the queue: flush_from_queue).
// Round 5, the scene's cadence at 1024^2 x 4 (roofline.scene_schedule): 144 Hz frames 110 -> 83 us per update, 60 Hz 137-142 -> 81, the heaviest
// frame 91 -> 61 us (144 Hz).
// how many of the cascades the ow_process calls will take (records[count - 1], [count - 2], ..) one launch can compute pass 1 for, 0 = none
int prearm_depth(const ow_context *c, const ow_cascade_params *records, int count) {
    constexpr int kRing = ow_context::Lookahead::kMaxAhead;
    if (lookahead_mode(c, 1) != 2) return 0;  // the group kernel's form only (a cascade of the layer-parallel compact family: <= 1 Mi texels)
    const int idx = count - 1;
    int depth = std::min(idx + 1, std::min(kRing, c->ahead_depth));
    for (int k = 0; k < depth; ++k) {
        const ow_cascade_params &p = records[idx - k];
        if (p.should_generate_spectrum || validate_record(p, idx - k) != OW_OK) depth = k;
    }
    return depth;
}
// does the head of the queue hold pass 1 of `cascade` as record p asks for it (one cascade per entry: the reference's schedule)?
bool queue_head_serves(const ow_context *c, int cascade, const ow_cascade_params &p) {
    const ow_context::Lookahead &la = c->la;
    if (!la.armed || la.queued < 1 || la.count != 1 || la.mode != 2) return false;
    const float t = (float)p.time;
    const int h = la.head;
    return la.cascade[h][0] == cascade && std::memcmp(&t, &la.time[h][0], 4) == 0 && p.tile_length[0] == la.tile_x[h][0] && p.tile_length[1] == la.tile_y[h][0];
}
// THE LEFTOVER RIDES WITH THE NEXT UPDATE'S PASS 1 (round 6).  On the scene's own cadence (144 Hz frames, an update every third frame, one ow_process per
// frame) every update finds ONE cascade of the previous arm unprocessed, its pass 1 waiting at the head of the queue, and used to launch its pass 2 alone
// (flush_from_queue: a quarter-filled launch) right in front of the pre-arm launch.  Both are inside this ow_update, independent of each other, and the
// group kernel takes exactly this shape -- pass 2 of one cascade (the layer-parallel family's block, what a batch of one takes anyway: bit-identical to
// the flush it replaces) beside pass 1 of up to four others, as every refilling ow_process launches it: `flush` = that leftover record, or nullptr.
// records: the armed records of this update.  Returns 1 = launched, 0 = nothing to launch (the ow_process calls take what they find), -1 = the launch failed.
int prearm_launch(ow_context *c, const ow_cascade_params *records, int count, const ow_cascade_params *flush) {
    ow_context::Lookahead &la = c->la;
    constexpr int kRing = ow_context::Lookahead::kMaxAhead;
    const int idx = count - 1;
    if (lookahead_mode(c, 1) != 2) return 0;  // the group kernel's form only; a context whose single cascades take the pair kernel (2048^2 x 1) keeps whatever
                                              // its last ow_process computed ahead for this update -- that call's own check decides
    if (!flush && la.armed && la.queued > 0) {
        // Work computed ahead is waiting already.  If its head is what the first ow_process of this update will ask for (a regular cadence: the previous
        // update's last ow_process guessed right), leave it.  If it cannot hit -- pair-kernel entries left by ow_update_all, a guess made with another
        // delta -- that call would miss, launch a lone pass 1 and refill: the round-4 path.  Drop the stale queue and pre-arm instead.
        if (queue_head_serves(c, idx, records[idx])) return 0;
        la.armed = false;
        la.queued = 0;
    }
    const int depth = prearm_depth(c, records, count);
    if (depth < 1) return 0;
    const int groups = kRing + 1, stride = 1;
    if (ensure_scratch(c, groups * stride) != OW_OK) return 0;  // (a pre-arm that would have to grow the scratch and cannot is simply not made; never with `flush`: its entry lives there)
    c->ra.armed = false;
    // the group that holds the launch's own pass-2 intermediate: the leftover's queue entry, or none (nothing is queued and everything launched so far
    // precedes this launch in stream order: the entries then sit in groups 1 .. depth, consecutive slots without a wrap, which is what flush_from_queue needs)
    const int cur = flush ? la.group[la.head] : 0;
    const int first1 = flush ? 1 : 0;
    ow::FrameArgs args;
    ow::TickGroupArgs ga;
    std::memset(&args, 0, sizeof(args));
    std::memset(&ga, 0, sizeof(ga));
    if (flush) {  // launch slot 0: pass 2 of cascade 0 of the previous arm, exactly as flush_from_queue would have launched it
        c->maps_faulted &= ~1u;
        c->enqueued_since_sync |= 1u;
        record_frame_constants(c, 0, *flush);
        args.c[0] = frame_of(*flush, 0);
        c->last_args = args;
        c->last_count = 1;
        c->last_family = ow::kernel_family(c->n, 1, c->kernel_mode);
        for (int &sl : c->slot_of) sl = -1;
        ga.d2 = 1;
        ga.tbase2[0] = cur * stride;
        ++la.hits;
    }
    la.head = 0;
    for (int k = 0; k < depth; ++k) {
        const ow_cascade_params &p = records[idx - k];
        la.group[k] = (cur + 1 + k) % groups;
        ga.tbase1[k] = la.group[k] * stride;
        args.c[first1 + k] = frame_of(p, idx - k);   // pass-1 "tick" k = launch slot first1 + k (step1 = 1): tile lengths and layer from it, the time from time1
        la.time[k][0] = ga.time1[k][first1] = (float)p.time;
        la.cascade[k][0] = idx - k;
        la.tile_x[k][0] = p.tile_length[0];
        la.tile_y[k][0] = p.tile_length[1];
    }
    ga.d1 = depth;
    ga.first1 = first1;
    ga.step1 = 1;
    ga.slots = 1;
    ga.p1_compact = c->group_p1_form >= 0 ? c->group_p1_form : ((size_t)depth * c->n * c->n >= ((size_t)2 << 20) ? 1 : 0);
    if (ow::launch_tick_group(c->n, args, ga, c->buf, main_stream(c)) != hipSuccess) {
        (void)hipGetLastError();
        la.armed = false;
        la.queued = 0;
        return -1;  // (without `flush` nothing is lost: the ow_process calls take the ordinary path -- and report the device's state themselves)
    }
    la.queued = depth;
    la.armed = true;
    la.count = 1;
    la.mode = 2;
    la.cur_group = cur;
    la.speculated += 1;
    return 1;
}

// The flush of ow_update (wave_generator.gd:94-98: the cascades 0 .. left - 1 the previous arm never got to) from the queue: if pass 1 of EVERY
// leftover is waiting there -- each entry checked exactly like a hit: cascade, FP32 time and tile lengths bit for bit, no spectrum to regenerate --
// and the entries sit in consecutive scratch slots, the flush is ONE pass-2 launch over them, in the kernel family a batch of `left` cascades takes
// anyway (launch_pass2 picks it from the batch size, as enqueue() would: the maps are bit-identical to the ordinary flush, whose pass 2 it is --
// pass 1 is the same lane code in every form and family, the families differ in pass 2).  Returns false when the ordinary flush has to do it.
bool flush_from_queue(ow_context *c, int left, ow_status *out) {
    ow_context::Lookahead &la = c->la;
    constexpr int kRing = ow_context::Lookahead::kMaxAhead;
    if (!la.armed || la.count != 1 || la.mode != 2 || left < 1 || left > kRing || la.queued < left) return false;
    if (lookahead_mode(c, 1) != 2 || batch_size(c, left) != left) return false;  // (timing, fault injection, no merging, pinned families: the ordinary path)
    const int g0 = la.group[la.head];
    for (int j = 0; j < left; ++j) {  // launch slot j = cascade left - 1 - j, the order ow_process would have taken them = the queue's order
        const int e = (la.head + j) % kRing, cascade = left - 1 - j;
        const ow_cascade_params &p = c->pass_parameters[cascade];
        const float t = (float)p.time;
        if (p.should_generate_spectrum || validate_record(p, cascade) != OW_OK || la.cascade[e][0] != cascade || std::memcmp(&t, &la.time[e][0], 4) != 0 ||
            p.tile_length[0] != la.tile_x[e][0] || p.tile_length[1] != la.tile_y[e][0] || la.group[e] != g0 + j)  // (consecutive slots: no wrap of the ring)
            return false;
    }
    ow::FrameArgs args;
    std::memset(&args, 0, sizeof(args));
    for (int j = 0; j < left; ++j) {
        const int cascade = left - 1 - j;
        const ow_cascade_params &p = c->pass_parameters[cascade];
        c->maps_faulted &= ~(1u << cascade);
        c->enqueued_since_sync |= 1u << cascade;
        record_frame_constants(c, cascade, p);
        args.c[j] = frame_of(p, cascade);
    }
    c->last_args = args;
    c->last_count = left;
    c->last_family = ow::kernel_family(c->n, left, c->kernel_mode);
    for (int &sl : c->slot_of) sl = -1;
    ow::DeviceBuffers b = c->buf;  // the batch's intermediate starts at scratch slot g0 (a queue entry of one cascade = one slot)
    const size_t pl = (size_t)c->n * c->n;
    b.T += (size_t)g0 * pl * ow::kLayers;
    b.pcol += (size_t)g0 * c->n;
    b.rrow += (size_t)g0 * c->n * 4;
    const hipError_t e = ow::launch_pass2(c->n, left, c->kernel_mode, args, b, main_stream(c));
    if (e != hipSuccess) {
        la.armed = false;
        la.queued = 0;
        *out = fail(OW_ERR_HIP, "flush from the look-ahead queue failed: %s", hipGetErrorString(e));
        return true;
    }
    la.hits += (uint64_t)left;
    la.head = (la.head + left) % kRing;
    la.queued -= left;
    la.armed = la.queued > 0;
    la.cur_group = g0 + left - 1;
    *out = OW_OK;
    return true;
}

ow_status check_cascade(const ow_context *c, int cascade) {
    if (!c) return fail(OW_ERR_INVALID, "null context");
    if (cascade < 0 || cascade >= c->layers) return fail(OW_ERR_INVALID, "cascade %d out of range [0,%d)", cascade, c->layers);
    return OW_OK;
}

}  // namespace

extern "C" {

const char *ow_last_error(void) { return g_last_error.c_str(); }
int32_t ow_abi_version(void) { return OW_ABI_VERSION; }

double ow_jonswap_alpha(double wind_speed, double fetch_length_m) {  // wave_generator.gd:116-117
    return 0.076 * std::pow(std::pow(wind_speed, 2.0) / (fetch_length_m * kHostG), 0.22);  // `wind_speed**2` is pow() in GDScript
}
double ow_jonswap_peak_angular_frequency(double wind_speed, double fetch_length_m) {  // wave_generator.gd:120-121
    return 22.0 * std::pow(kHostG * kHostG / (wind_speed * fetch_length_m), 1.0 / 3.0);
}

static_assert(sizeof(ow_cascade_params) == 128 && offsetof(ow_cascade_params, displacement_scale) == 8 && offsetof(ow_cascade_params, spectrum_seed) == 88 &&
                  offsetof(ow_cascade_params, time) == 104, "ow_cascade_params layout (ABI 4): the C#, ctypes and C++ mirrors are written against it");
void ow_cascade_params_default(ow_cascade_params *p) {  // wave_cascade_parameters.gd:7-42
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->tile_length[0] = p->tile_length[1] = 50.0f;
    p->displacement_scale = 1.0;
    p->normal_scale = 1.0;
    p->wind_speed = 20.0;
    p->wind_direction = 0.0;
    p->fetch_length = 550.0;
    p->swell = 0.8;
    p->spread = 0.2;
    p->detail = 1.0;
    p->whitecap = 0.5;
    p->foam_amount = 5.0;
    p->should_generate_spectrum = 1;
}

ow_status ow_create(const ow_config *cfg, ow_context **out) {
    if (!cfg || !out) return fail(OW_ERR_INVALID, "null argument");
    *out = nullptr;
    if (!ow::supported_map_size(cfg->map_size))
        return fail(OW_ERR_INVALID, "map_size %d unsupported (128, 256, 512, 1024, 2048)", cfg->map_size);
    if (cfg->num_cascades < 1 || cfg->num_cascades > OW_MAX_CASCADES)
        return fail(OW_ERR_INVALID, "num_cascades %d outside [1,%d]", cfg->num_cascades, OW_MAX_CASCADES);

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(OW_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    int dev = cfg->device_id, caller_dev = 0;
    OW_HIP(hipGetDevice(&caller_dev));
    if (dev < 0) dev = caller_dev;
    if (dev >= ndev) return fail(OW_ERR_INVALID, "device_id %d >= device count %d", dev, ndev);
    // the caller's current device is put back on every way out (a library must not change it behind the caller's back)
    struct DeviceRestore {
        int dev;
        ~DeviceRestore() { (void)hipSetDevice(dev); }
    } restore{caller_dev};
    OW_HIP(hipSetDevice(dev));
    hipDeviceProp_t prop;
    OW_HIP(hipGetDeviceProperties(&prop, dev));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(OW_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", dev, prop.gcnArchName);

    ow_context *c = new (std::nothrow) ow_context();
    if (!c) return fail(OW_ERR_NOMEM, "out of host memory");
    c->n = cfg->map_size;
    c->cascades = cfg->num_cascades;
    c->layers = cfg->num_cascades < 2 ? 2 : cfg->num_cascades;  // init_gpu(maxi(2, n)), water.gd:91
    c->device = dev;
    c->depth = cfg->depth > 0.0f ? cfg->depth : 20.0f;  // DEPTH, wave_generator.gd:6
    {
        const bool lp = cfg->flags & OW_FLAG_KERNELS_LAYER_PARALLEL, cp = cfg->flags & OW_FLAG_KERNELS_COMPACT;
        c->kernel_mode = (cfg->flags & OW_FLAG_KERNELS_STANDARD) ? 1 : (lp && cp) ? 4 : lp ? 2 : cp ? 3 : 0;
    }

    auto bail = [&](ow_status st) {
        ow_destroy(c);
        return st;
    };
    if (cfg->stream) {
        c->stream = (hipStream_t)cfg->stream;
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess)
            return bail(fail(OW_ERR_HIP, "hipStreamCreate failed"));
        c->own_stream = true;
    }
    // the second chain's stream (ow_kernels.h "TWO CHAINS"): only where a launch can be split at all -- 1024^2 with at least four cascades, 512^2 with eight
    if (!(cfg->flags & OW_FLAG_SINGLE_STREAM) && ((c->n == 1024 && c->cascades >= 4) || (c->n == 512 && c->cascades >= 8))) {
        if (hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->side_fork_ev, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->side_join_ev, hipEventDisableTiming) != hipSuccess)
            return bail(fail(OW_ERR_HIP, "second stream / events: creation failed"));
    }
    const size_t pl = plane(c), L = (size_t)c->layers;
#define OW_ALLOC(ptr, bytes)                                                                              \
    if (hipMalloc((void **)&(ptr), (bytes)) != hipSuccess)                                                \
        return bail(fail(OW_ERR_NOMEM, "hipMalloc of %zu bytes failed for " #ptr, (size_t)(bytes)));
    OW_ALLOC(c->buf.h0, L * pl * sizeof(ow::cplx));                   // h0(k): the non-redundant half of the spectrum texture (:31)
    OW_ALLOC(c->buf.omega, L * pl * sizeof(float));
    // scratch between pass 1 and pass 2 of one batch (half of the reference's fft_buffer, :33): one batch worth only, so it
    // is the same <= 128 MiB for every batch and stays in the Infinity Cache
    c->no_merge = (cfg->flags & OW_FLAG_NO_TICK_GROUPS) != 0;
    c->run_as_calls = (cfg->flags & OW_FLAG_RUN_AS_CALLS) != 0;
    c->run_as_reference = (cfg->flags & OW_FLAG_RUN_AS_REFERENCE_SCHEDULE) != 0;
    c->always_regenerate = (cfg->flags & OW_FLAG_ALWAYS_REGENERATE_SPECTRUM) != 0;
    plan_tick_groups(c, cfg->flags);
    // (OW_FLAG_LAZY_SCRATCH: one batch now, the look-ahead's share on its first use -- a context that is only ever driven through ow_run's tick groups,
    //  or one of many shards on a device, need not hold it)
    const bool lazy_scratch = (cfg->flags & OW_FLAG_LAZY_SCRATCH) != 0;
    if (ensure_scratch(c, lazy_scratch ? base_scratch_slots(c) : std::max(base_scratch_slots(c), lookahead_scratch_slots(c))) != OW_OK) return bail(OW_ERR_NOMEM);
    if (cfg->displacement_map) {
        c->buf.disp = (ow::u16x4 *)cfg->displacement_map;
    } else {
        OW_ALLOC(c->buf.disp, exportable_bytes(L * pl * sizeof(ow::u16x4)));  // R16G16B16A16_SFLOAT (:34)
        c->own_disp = true;
    }
    if (cfg->normal_map) {
        c->buf.norm = (ow::u16x4 *)cfg->normal_map;
    } else {
        OW_ALLOC(c->buf.norm, exportable_bytes(L * pl * sizeof(ow::u16x4)));  // (:35)
        c->own_norm = true;
    }
    OW_ALLOC(c->buf.foam, L * pl * sizeof(uint16_t));                 // FP16 foam state in pass-2 lane order
    if (cfg->flags & OW_FLAG_DEBUG_F32) { OW_ALLOC(c->buf.f32, L * pl * 8 * sizeof(float)); }
    if (hipHostMalloc((void **)&c->status_host, 64, hipHostMallocMapped) != hipSuccess)
        return bail(fail(OW_ERR_NOMEM, "hipHostMalloc failed for the device status word"));
    *c->status_host = 0u;
    if (hipHostGetDevicePointer((void **)&c->buf.status, c->status_host, 0) != hipSuccess)
        return bail(fail(OW_ERR_HIP, "hipHostGetDevicePointer failed for the device status word"));
    std::vector<ow::cplx> tw;
    ow::make_twiddles(c->n, tw);
    OW_ALLOC(c->tw_dev, tw.size() * sizeof(ow::cplx));
    std::vector<ow::cplx> tw_split;
    if (ow::make_split_twiddles(c->n, tw_split)) { OW_ALLOC(c->tw_split_dev, tw_split.size() * sizeof(ow::cplx)); }
    std::vector<ow::cplx> tw_half;
    if (ow::make_half_twiddles(c->n, tw_half)) { OW_ALLOC(c->tw_half_dev, tw_half.size() * sizeof(ow::cplx)); }
#undef OW_ALLOC
    c->buf.tw = c->tw_dev;
    c->buf.tw_split = c->tw_split_dev;
    c->buf.tw_half = c->tw_half_dev;
    if (c->tw_half_dev && hipMemcpyAsync(c->tw_half_dev, tw_half.data(), tw_half.size() * sizeof(ow::cplx), hipMemcpyHostToDevice, main_stream(c)) != hipSuccess)
        return bail(fail(OW_ERR_HIP, "twiddle upload failed"));
    if (c->tw_split_dev && hipMemcpyAsync(c->tw_split_dev, tw_split.data(), tw_split.size() * sizeof(ow::cplx), hipMemcpyHostToDevice, main_stream(c)) != hipSuccess)
        return bail(fail(OW_ERR_HIP, "twiddle upload failed"));
    // Vulkan images start undefined; foam must start from a defined state: zero
    if (hipMemcpyAsync(c->tw_dev, tw.data(), tw.size() * sizeof(ow::cplx), hipMemcpyHostToDevice, main_stream(c)) != hipSuccess ||
        hipMemsetAsync(c->buf.disp, 0, L * pl * sizeof(ow::u16x4), main_stream(c)) != hipSuccess ||
        hipMemsetAsync(c->buf.norm, 0, L * pl * sizeof(ow::u16x4), main_stream(c)) != hipSuccess ||
        hipMemsetAsync(c->buf.foam, 0, L * pl * sizeof(uint16_t), main_stream(c)) != hipSuccess ||
        hipMemsetAsync(c->buf.h0, 0, L * pl * sizeof(ow::cplx), main_stream(c)) != hipSuccess ||
        hipMemsetAsync(c->buf.omega, 0, L * pl * sizeof(float), main_stream(c)) != hipSuccess ||
        hipStreamSynchronize(main_stream(c)) != hipSuccess)
        return bail(fail(OW_ERR_HIP, "initial upload failed: %s", hipGetErrorString(hipGetLastError())));
    for (int &sl : c->slot_of) sl = -1;
    *out = c;
    return OW_OK;
}

void ow_destroy(ow_context *c) {
    if (!c) return;
    int caller_dev = -1;
    (void)hipGetDevice(&caller_dev);
    (void)hipSetDevice(c->device);
    if (c->side_stream) (void)hipStreamSynchronize(c->side_stream);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->buf.h0);
    (void)hipFree(c->buf.omega);
    (void)hipFree(c->buf.T);
    if (c->own_disp) (void)hipFree(c->buf.disp);
    if (c->own_norm) (void)hipFree(c->buf.norm);
    (void)hipFree(c->buf.foam);
    (void)hipFree(c->buf.f32);
    (void)hipFree(c->buf.pcol);
    (void)hipFree(c->buf.rrow);
    (void)hipFree(c->tw_dev);
    (void)hipFree(c->tw_split_dev);
    (void)hipFree(c->tw_half_dev);
    if (c->status_host) (void)hipHostFree(c->status_host);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    (void)hipFree(c->snap_dev);
    if (c->snap_host) (void)hipHostFree(c->snap_host);
    for (auto &e : c->snap_ready)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : c->copy_done)
        if (e) (void)hipEventDestroy(e);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    (void)hipFree(c->query_xz);
    (void)hipFree(c->query_out);
    for (auto &e : c->ev)
        if (e) (void)hipEventDestroy(e);
    if (c->side_fork_ev) (void)hipEventDestroy(c->side_fork_ev);
    if (c->side_join_ev) (void)hipEventDestroy(c->side_join_ev);
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    if (caller_dev >= 0) (void)hipSetDevice(caller_dev);
}

namespace {
ow_status update_impl(ow_context *c, double delta, ow_cascade_params *params, int32_t count, bool process_calls_follow);
// The caller's cadence (look-ahead: lookahead_tick / lookahead_process).  "The same delta" tolerates a nanosecond: a fixed-step scene behind
// water.gd's rate limiter (:78, target + (time - next_update_time)) issues deltas that are equal up to the rounding noise of its FP64 clock,
// and what has to repeat for a hit is the FP32-narrowed TIME (ulp 7.6e-6 s from t = 64 s on), which the hit check compares bit for bit anyway.
// (measured against the delta that STARTED the streak, so that drift does not accumulate unnoticed, and relative for large deltas.)
void note_cadence(ow_context *c, double delta) {
    if (std::fabs(delta - c->la.streak_delta) <= 1e-9 * std::max(1.0, std::fabs(c->la.streak_delta))) {
        if (c->la.streak < (1 << 30)) ++c->la.streak;
    } else {
        c->la.prev_run = c->la.streak + 1;
        c->la.streak = 0;
        c->la.streak_delta = delta;
    }
    c->la.last_delta = delta;
}
}
ow_status ow_update(ow_context *c, double delta, ow_cascade_params *params, int32_t count) { return update_impl(c, delta, params, count, true); }
namespace {
ow_status update_impl(ow_context *c, double delta, ow_cascade_params *params, int32_t count, bool process_calls_follow) {
    if (!c || !params) return fail(OW_ERR_INVALID, "null argument");
    if (count < 1 || count > c->cascades)  // assert(parameters.size() != 0), :91
        return fail(OW_ERR_INVALID, "count %d outside [1,%d]", count, c->cascades);
    if (!std::isfinite(delta)) return fail(OW_ERR_INVALID, "delta is not finite");
    // the new records are checked BEFORE anything changes: a refused call has advanced no time, consumed no dirty flag, armed nothing
    if (ow_status st = ow::validate_records(params, count, delta); st != OW_OK) return st;
    OW_HIP(hipSetDevice(c->device));
    // what this update arms, computed aside first (:101-106; GDScript floats are FP64): nothing of `params` or of the context changes before the
    // leftovers' flush has been launched -- a call that fails there can simply be repeated
    ow_cascade_params next[OW_MAX_CASCADES], armed[OW_MAX_CASCADES];
    for (int i = 0; i < count; ++i) {
        ow_cascade_params &p = next[i];
        p = params[i];
        p.time += delta;
        p.foam_grow_rate = delta * p.foam_amount * 7.5;
        const double d = 10.0 - p.foam_amount;
        p.foam_decay_rate = delta * (d > 0.5 ? d : 0.5) * 1.15;
        armed[i] = p;                     // :108 -- a copy: `params` is not touched after this call returns
        p.should_generate_spectrum = 0;   // consumed: the armed copy carries it until the cascade is processed (:72)
    }
    // ... unless the spectrum it asks for is the one that is there (spectrum_is_resident) -- asked AFTER the flush, which may itself regenerate a layer
    auto settle_armed = [&]() {
        for (int i = 0; i < count; ++i) settle_dirty_flag(c, i, armed[i]);
    };
    bool prearmed = false;
    if (c->pass_num_cascades_remaining != 0) {  // :94-98: leftovers of the previous arm, with the previous records
        int idx[OW_MAX_CASCADES];
        const int left = c->pass_num_cascades_remaining;
        for (int i = 0; i < left; ++i) idx[i] = i;
        // Whatever happens to them, the leftovers do not survive this call: were a failed flush to stay armed, every later
        // ow_update would run into it again.  (Armed records were validated on the way in, so only a HIP failure can end up here;
        // nothing of `params` has been touched yet, so the call can simply be repeated.)
        c->pass_num_cascades_remaining = 0;
        ow_status st = OW_OK;
        const ow_cascade_params leftover = c->pass_parameters[0];
        // ONE leftover whose pass 1 waits alone at the head of the queue, and an update whose cascades can be pre-armed: its pass 2 rides in the pre-arm launch
        // (the leftover has no spectrum to regenerate, so the resident spectra the new records are settled against are already the ones they will meet)
        if (process_calls_follow && left == 1 && c->la.queued == 1 && !leftover.should_generate_spectrum && validate_record(leftover, 0) == OW_OK &&
            lookahead_mode(c, 1) == 2 && batch_size(c, 1) == 1 && queue_head_serves(c, 0, leftover) && (settle_armed(), prearm_depth(c, armed, count) >= 1)) {
            const int r = prearm_launch(c, armed, count, &leftover);
            if (r < 0) return fail(OW_ERR_HIP, "the launch that flushes the previous update's leftover failed");
            prearmed = r > 0;
        }
        // (otherwise their pass 1 may still be waiting in the queue, computed ahead for the ow_process calls that never came: then the flush is one pass-2 launch)
        if (!prearmed && !flush_from_queue(c, left, &st)) st = enqueue(c, c->pass_parameters, idx, left);
        if (st != OW_OK) return st;
    }
    settle_armed();
    note_cadence(c, delta);
    if (!c->inside_run) c->ra.last_was_run = false;  // (a tick issued by the caller itself: the next ow_run does not "follow a run")
    for (int i = 0; i < count; ++i) {
        params[i] = next[i];
        c->pass_parameters[i] = armed[i];
    }
    c->pass_count = count;
    c->pass_num_cascades_remaining = count;  // :109
    if (process_calls_follow && !prearmed) (void)prearm_launch(c, c->pass_parameters, count, nullptr);
    return OW_OK;
}
}  // namespace

ow_status ow_set_cascade_params(ow_context *c, int32_t index, const ow_cascade_params *p) {
    if (!c || !p) return fail(OW_ERR_INVALID, "null argument");
    if (index < 0 || index >= c->pass_count) return fail(OW_ERR_INVALID, "index %d outside the %d records of the last ow_update", index, c->pass_count);
    if (ow_status st = validate_record(*p, index); st != OW_OK) return st;  // a refused record leaves the armed copy as it was
    c->pass_parameters[index] = *p;
    settle_dirty_flag(c, index, c->pass_parameters[index]);
    return OW_OK;
}

ow_status ow_get_cascade_params(const ow_context *c, int32_t index, ow_cascade_params *out) {
    if (!c || !out) return fail(OW_ERR_INVALID, "null argument");
    if (index < 0 || index >= c->pass_count) return fail(OW_ERR_INVALID, "index %d outside the %d records of the last ow_update", index, c->pass_count);
    *out = c->pass_parameters[index];
    return OW_OK;
}

ow_status ow_process(ow_context *c) {  // :56-63
    if (!c) return fail(OW_ERR_INVALID, "null context");
    if (c->pass_num_cascades_remaining == 0) return OW_OK;
    OW_HIP(hipSetDevice(c->device));
    if (!c->inside_run) c->ra.last_was_run = false;
    const int idx = c->pass_num_cascades_remaining - 1;
    ow_status st = OW_OK;
    if (!lookahead_process(c, idx, &st)) st = enqueue(c, c->pass_parameters, &idx, 1);
    if (st != OW_OK) return st;               // a cascade that could not be enqueued stays armed
    c->pass_num_cascades_remaining -= 1;
    return OW_OK;
}

ow_status ow_update_all(ow_context *c, double delta, ow_cascade_params *params, int32_t count) {
    ow_status st = update_impl(c, delta, params, count, false);
    if (st != OW_OK) return st;
    int idx[OW_MAX_CASCADES];
    for (int i = 0; i < count; ++i) idx[i] = count - 1 - i;  // same order _process would take
    if (!lookahead_tick(c, delta, count, &st)) st = enqueue(c, c->pass_parameters, idx, count);
    if (st != OW_OK) return st;
    c->pass_num_cascades_remaining = 0;
    if (!c->inside_run) join_for_caller(c);
    return OW_OK;
}

ow_status ow_chain_stats(const ow_context *c, uint64_t *split_launches) {
    if (!c) return fail(OW_ERR_INVALID, "null context");
    if (split_launches) *split_launches = c->split_launches;
    return OW_OK;
}

ow_status ow_lookahead_stats(const ow_context *c, uint64_t *hits, uint64_t *speculated) {
    if (!c) return fail(OW_ERR_INVALID, "null context");
    if (hits) *hits = c->la.hits;
    if (speculated) *speculated = c->la.speculated;
    return OW_OK;
}

ow_status ow_spectrum_stats(const ow_context *c, uint64_t *generated, uint64_t *skipped) {
    if (!c) return fail(OW_ERR_INVALID, "null context");
    if (generated) *generated = c->spectra_generated;
    if (skipped) *skipped = c->spectra_skipped;
    return OW_OK;
}

ow_status ow_debug_inject_fault(ow_context *c, uint32_t fault_bits) {
    if (!c) return fail(OW_ERR_INVALID, "null context");
    if (fault_bits & 2u) {  // the status word as a faulting launch would leave it -- of the launches IN FLIGHT (look-ahead included), not of the next batch
        if (c->status_host) __atomic_store_n(c->status_host, (uint32_t)ow::kStatusRowSyncTimeout, __ATOMIC_RELEASE);
        fault_bits &= ~2u;
        if (fault_bits == 0) return OW_OK;
    }
    c->inject_fault = fault_bits;
    return OW_OK;
}

namespace {

// Can the remaining ticks of ow_run go out as tick groups / tick pairs?  Only a batch of the layer-parallel compact family (groups)
// or a batch of the compact family of at most kPairTexels (pairs), with nothing left
// armed, no spectrum to regenerate, no fault to inject, no per-launch timing requested, and records that pass enqueue()'s checks.
// returns 0 = not usable, the ticks per launch (group_depth) for the tick groups, -1 for the compact family's tick pairs
int tick_groups_usable(const ow_context *c, const ow_cascade_params *params, int count) {
    if (c->timing == 1 || c->inject_fault || c->pass_num_cascades_remaining != 0) return 0;
    int sizes[OW_MAX_CASCADES];
    const bool groups = ow::kernel_family(c->n, count, c->kernel_mode) == 4 && count <= c->group_max_count;
    const bool pairs = !groups && c->pair_slots > 0 && pair_batches(c, count, sizes) > 0;
    if (!groups && !pairs) return 0;
    for (int i = 0; i < count; ++i)
        if (params[i].should_generate_spectrum || !finite_record(params[i]) || !(params[i].tile_length[0] > 0.0f) || !(params[i].tile_length[1] > 0.0f))
            return 0;  // (the ordinary path regenerates / reports)
    return groups ? tick_group_depth_for(c, count) : -1;
}

// one more ow_update_all() worth of arithmetic on the records (wave_generator.gd:101-106); time_out[i] = FP32 time of launch
// slot i (= cascade count-1-i, as in ow_update_all)
void advance_tick(double delta, ow_cascade_params *params, int count, float *time_out) {
    for (int i = 0; i < count; ++i) {
        ow_cascade_params &p = params[count - 1 - i];
        p.time += delta;
        p.foam_grow_rate = delta * p.foam_amount * 7.5;
        const double d = 10.0 - p.foam_amount;
        p.foam_decay_rate = delta * (d > 0.5 ? d : 0.5) * 1.15;
        time_out[i] = (float)p.time;
    }
}

// the launch constants that do not change from tick to tick of a run (same delta: same foam rates), launch slot i = cascade count-1-i
ow::FrameArgs run_frame_args(const ow_cascade_params *params, int count) {
    ow::FrameArgs args;
    std::memset(&args, 0, sizeof(args));
    for (int i = 0; i < count; ++i) {
        const ow_cascade_params &p = params[count - 1 - i];
        ow::CascadeFrame &cf = args.c[i];
        cf.tile_x = p.tile_length[0];
        cf.tile_y = p.tile_length[1];
        cf.whitecap = (float)p.whitecap;
        cf.foam_grow_rate = (float)p.foam_grow_rate;
        cf.foam_decay = expf(-(float)p.foam_decay_rate);
        cf.cascade = count - 1 - i;
    }
    return args;
}
ow_status launch_merged(ow_context *c, const ow::FrameArgs &args, const ow::TickGroupArgs &ga) {
    hipEvent_t *ev = nullptr;
    if (c->timing == 2) {  // as-launched timing: this launch's own begin -> end
        ow_status st = next_events(c, &ev, true);
        if (st != OW_OK) return st;
    }
    OW_HIP(launch_group(c, args, ga, ow::LaunchTiming{ev ? ev[0] : nullptr, ev ? ev[1] : nullptr}));
    return OW_OK;
}
void finish_merged_run(ow_context *c, ow::FrameArgs &args, const ow_cascade_params *params, int count, int last_batch, int family, int depth) {
    for (int i = 0; i < count; ++i) {
        c->pass_parameters[i] = params[i];
        args.c[i].time = (float)params[count - 1 - i].time;
        record_frame_constants(c, i, params[i]);  // (of the run's last tick)
    }
    c->pass_count = count;
    c->pass_num_cascades_remaining = 0;
    c->maps_faulted &= ~((1u << count) - 1u);  // (as enqueue(): the run recomputed layers 0 .. count - 1)
    c->enqueued_since_sync |= (1u << count) - 1u;
    c->last_args = args;
    c->last_count = last_batch;
    c->last_family = family;
    c->last_group_depth = depth;
    for (int &sl : c->slot_of) sl = -1;  // (no reference-layout intermediate to inspect after such a run)
}

// RUNS THAT FOLLOW RUNS (round 6).  A host that calls ow_run again and again -- bench.py's timed regions, a headless simulation in chunks -- used to
// pay, per call, an ordinary first tick (flush, spectra, validation: two launches per batch, nothing merged) and two half-filled launches at the ends
// of the merged stream (pass 1 of the first group / batch alone, pass 2 of the last alone): 195 us at 1024^2 x 8, 268 at 2048^2 x 4, 42 at 256^2 x 4
// (profiles/r05_run_overhead.txt).  Now the LAST launch of a run that itself followed a run with the same delta and cascade count also carries pass 1
// of what the NEXT such run would start with -- the first tick group (times + delta, + 2 delta, ..: as many ticks as this run's first group had) or,
// in the cascade-major pair stream, the next tick of the batch the run ended on -- and the next run checks it exactly like a look-ahead hit (count,
// every FP32 time and tile length bit for bit, nothing armed, no spectrum to generate, nothing else has touched the scratch) and starts in the middle
// of the stream: every launch of back-to-back runs is a full one.  A miss costs what a run cost before.  The first run after anything else never
// speculates, so a one-shot caller (one ow_run, then a readback) leaves nothing in the queue.  Bit-identical either way (the same item bodies).
// The pair stream also turns round at every block: batch order 0 .. B-1, then B-1 .. 0, ..., across runs as well (pair_dir) -- the batch that ends a
// block (a run) is the one the next starts with, still warm in the Infinity Cache (round 5's trace: the first launch of a 20-tick run at 1024^2 x 8,
// pass 1 of the batch the previous run had left cold, 55 us instead of 28).
// 0 = the run starts the ordinary way; otherwise what tick_groups_usable returns, and c->ra holds pass 1 of this run's first group / batch
int run_resume_usable(ow_context *c, double delta, const ow_cascade_params *params, int count, int frames) {
    const ow_context::RunAhead &ra = c->ra;
    if (!ra.armed || ra.count != count || frames < 1 || c->timing != 0 || c->inject_fault || c->pass_num_cascades_remaining != 0) return 0;
    if (ow::validate_records(params, count, delta) != OW_OK) return 0;  // (the ordinary path reports it)
    ow_cascade_params tmp[OW_MAX_CASCADES];
    for (int i = 0; i < count; ++i) {
        tmp[i] = params[i];
        if (tmp[i].should_generate_spectrum) {  // a dirty record whose spectrum is resident is as good as a clean one (spectrum_is_resident)
            if (!spectrum_is_resident(c, i, tmp[i])) return 0;
            tmp[i].should_generate_spectrum = 0;
        }
    }
    const int depth = tick_groups_usable(c, tmp, count);
    if (depth == 0 || (depth > 0) != (ra.kind == 1)) return 0;
    for (int i = 0; i < count; ++i) {  // launch slot i = cascade count - 1 - i
        const ow_cascade_params &p = tmp[count - 1 - i];
        if (p.tile_length[0] != ra.tile_x[i] || p.tile_length[1] != ra.tile_y[i]) return 0;
    }
    auto time_is = [&](int slot, int adds, float expect) {
        double t = tmp[count - 1 - slot].time;
        for (int j = 0; j < adds; ++j) t += delta;  // wave_generator.gd:103: one FP64 add per update, narrowed by the pack
        const float ft = (float)t;
        return std::memcmp(&ft, &expect, 4) == 0;
    };
    if (ra.kind == 1) {
        if (depth != ra.D || ra.ticks < 1) return 0;
        const int use = std::min(ra.ticks, frames);
        for (int j = 0; j < use; ++j)
            for (int i = 0; i < count; ++i)
                if (!time_is(i, j + 1, ra.time[j][i])) return 0;
    } else {
        int sizes[OW_MAX_CASCADES];
        const int B = pair_batches(c, count, sizes);
        if (B < 1 || ra.batch != (c->pair_dir ? B - 1 : 0) || ra.batch >= B) return 0;
        int first = 0;
        for (int b = 0; b < ra.batch; ++b) first += sizes[b];
        if (first != ra.first || sizes[ra.batch] != ra.size) return 0;
        for (int i = first; i < first + ra.size; ++i)
            if (!time_is(i, 1, ra.time[0][i])) return 0;
    }
    return depth;
}

// `ticks` >= 1 consecutive ow_update_all() ticks in groups of D = group_depth:
//   [pass 1 of group 0] [pass 2 of group 0 + pass 1 of group 1] ... [pass 2 of the last group (+ pass 1 of the next run's first group)]
// The scratch is a ring of 2 D ticks (tick at ring position r lives in slots (r mod 2D) * count ...); a launch reads one group and writes the next,
// each at most D ticks.  resume: group 0's pass 1 is in the ring already (c->ra: the previous run's last launch), no first launch.
ow_status run_tick_groups(ow_context *c, double delta, ow_cascade_params *params, int count, int ticks, int D, bool resume, bool speculate) {
    ow_context::RunAhead &ra = c->ra;
    c->la.armed = false;  // (the run uses the whole scratch its own way)
    c->la.queued = 0;
    const int ring = 2 * D;
    const int pos = resume ? ra.pos : 0;
    const int g0 = resume ? std::min(ra.ticks, ticks) : std::min(D, ticks);
    ra.armed = false;  // consumed (whatever of it this run does not use is dropped)
    auto slot_of = [&](int t) { return ((pos + t) % ring) * count; };
    ow::TickGroupArgs ga;
    std::memset(&ga, 0, sizeof(ga));
    ga.slots = count;
    // pass-1 items: k_pass1c's form (8 rows, all layers from one load + modulation) for maps of 512^2 up and from 384 Ki texels per
    // tick on, the layer-parallel form (more, smaller items; the spectrum is modulated once per layer) below -- measured, MI355X, us
    // per tick lp / compact: 256^2 x 1 4.40 / 4.50, x 4 5.02 / 5.05, x 5 6.01 / 6.28, x 6 7.06 / 6.70, x 8 9.55 / 7.77;
    // 512^2 x 1 7.25 / 7.04, x 2 10.1 / 8.5, x 4 19.4 / 15.2, x 6 31.4 / 28.1; 1024^2 x 1 19.0 / 15.8   (scripts/group_p1_body.py)
    ga.p1_compact = c->group_p1_form >= 0 ? c->group_p1_form : (c->n >= 512 || (size_t)count * c->n * c->n >= ((size_t)384 << 10));
    // pass-2 blocks: the pipelined form (k_tick_group_c_lp<.., PIPE>: a block's two halves on alternate ticks) while its blocks are at most
    // one per CU -- there a launch lasts as long as the chain of ticks through one wave, and the chain is what the pipeline shortens;
    // with it the pass-1 items take k_pass1c's form (fewer blocks beside the resident pass-2 blocks).  Measured, MI355X, us per tick
    // plain / pipelined (scripts/group_p2_form.py): 256^2 x 1 4.32 / 2.96, x 2 4.45 / 3.04, x 4 4.84 / 3.70, x 8 7.56 / 9.14;
    // 512^2 x 1 5.86 / 5.47, x 2 7.74 / 9.97, x 4 14.2 / 14.5
    const bool pipe_fits = ow::tick_group_pipe_blocks(c->n, count) > 0 && ow::tick_group_pipe_blocks(c->n, count) <= 256;
    ga.p2_pipe = c->group_p2_form >= 0 ? c->group_p2_form : pipe_fits;
    if (ga.p2_pipe && c->group_p1_form < 0) ga.p1_compact = 1;
    // group 0's ticks: the records advance through them either way; their pass 1 is launched here, or has been by the previous run
    ga.d2 = 0;
    ga.d1 = g0;
    for (int j = 0; j < g0; ++j) {
        advance_tick(delta, params, count, ga.time1[j]);
        ga.tbase1[j] = slot_of(j);
    }
    ow::FrameArgs args = run_frame_args(params, count);
    if (resume) {
        c->la.hits += (uint64_t)g0;
    } else if (ow_status st = launch_merged(c, args, ga); st != OW_OK) {
        return st;
    }
    for (int done = 0, size = g0; done < ticks;) {
        ga.d2 = size;
        for (int j = 0; j < size; ++j) ga.tbase2[j] = slot_of(done + j);
        const int next = std::min(D, ticks - done - size);
        bool arming = false;
        ga.d1 = next;
        for (int j = 0; j < next; ++j) {
            advance_tick(delta, params, count, ga.time1[j]);
            ga.tbase1[j] = slot_of(done + size + j);
        }
        if (next == 0 && speculate) {  // the run's last launch: pass 1 of the first group of a run like this one, should it follow
            ow_cascade_params ahead[OW_MAX_CASCADES];
            for (int i = 0; i < count; ++i) ahead[i] = params[i];
            const int S = std::min(D, ticks);
            ga.d1 = S;
            for (int j = 0; j < S; ++j) {
                advance_tick(delta, ahead, count, ga.time1[j]);
                ga.tbase1[j] = slot_of(ticks + j);
                for (int i = 0; i < count; ++i) ra.time[j][i] = ga.time1[j][i];
            }
            for (int i = 0; i < count; ++i) {
                ra.tile_x[i] = params[count - 1 - i].tile_length[0];
                ra.tile_y[i] = params[count - 1 - i].tile_length[1];
            }
            ra.kind = 1, ra.count = count, ra.D = D, ra.ticks = S, ra.pos = (pos + ticks) % ring;
            arming = true;
        }
        if (ow_status st = launch_merged(c, args, ga); st != OW_OK) return st;
        if (arming) {
            ra.armed = true;
            ++c->la.speculated;
        }
        done += size;
        size = next;
    }
    finish_merged_run(c, args, params, count, count, 5, D);
    return OW_OK;
}

// `ticks` >= 1 consecutive ow_update_all() ticks of the compact family as tick pairs: the run is a stream of batches (B per tick, launch
// slots in the order ow_update_all takes them), launch i = [pass 2 of batch i + pass 1 of batch i + 1]; the intermediates alternate between the two
// halves of the scratch (pair_slots each).
// ORDER OF THE STREAM.  Cascades are independent and a batch's ticks only have to follow each other, so any interleaving of the batches'
// tick sequences leaves the same bits behind.  A tick of several batches goes CASCADE-major in blocks of kPairTickBlock ticks: batch 0 through
// 64 ticks, then batch 1 through the same 64, ... -- every launch pairs pass 2 of a batch with pass 1 of THE SAME batch one tick later, so
//  * 63 launches out of 64 find their spectra (read by the previous launch), their foam (written by it) and both intermediates in the 256 MiB
//    Infinity Cache, exactly like a one-batch run -- at 2048^2 (one cascade per batch, 48 MB of spectra + 16 MB of foam each) tick-major order
//    re-reads everything from DRAM every tick and LOSES to one launch per pass at x 2 .. x 4;
//  * the two passes of a launch always cover the same number of cascades (1024^2 x 5 = 3 + 2: in tick-major order every launch pairs unequal
//    batches and the surplus blocks run unpaired).
// Measured (us per tick, one launch per pass | pairs tick-major | cascade-major x 64; profiles/r04_2048_pairs.txt, r04_pairs_order_1024.txt):
// 2048^2 x 2 123.8 | 140.2 | 117.2;  x 4 260.7 | 275.2 | 235.5;  x 8 554.9 | 520.5 | 471.9;  1024^2 x 5 76.4 | 75.8 | 68.5;  1024^2 x 8 110.1 |
// 108.4 (half-size batches) | 104.9 (full-size) -- bit-identical maps.  Only the final state of a run is defined for a caller (ow_run =
// `frames` ow_update_all calls back to back), and it is the same.  The blocks alternate in direction (0 .. B-1, then B-1 .. 0: round 6, see above).
constexpr int kPairTickBlock = 64;
void advance_slots(double delta, ow_cascade_params *params, int count, int first_slot, int nslots, float *time_out) {  // launch slot i = cascade count-1-i
    for (int i = first_slot; i < first_slot + nslots; ++i) {
        ow_cascade_params &p = params[count - 1 - i];
        p.time += delta;  // wave_generator.gd:103, one FP64 add per tick, in order
        time_out[i] = (float)p.time;
    }
}
ow_status run_tick_pairs(ow_context *c, double delta, ow_cascade_params *params, int count, int ticks, bool resume, bool speculate) {
    ow_context::RunAhead &ra = c->ra;
    c->la.armed = false;  // (the run uses the whole scratch its own way)
    c->la.queued = 0;
    int sizes[OW_MAX_CASCADES], first[OW_MAX_CASCADES];
    const int B = pair_batches(c, count, sizes);
    for (int b = 0, at = 0; b < B; ++b) {
        first[b] = at;
        at += sizes[b];
    }
    // the stream: which batch each launch's pass 2 belongs to
    const int block = c->pair_tick_block > 0 ? c->pair_tick_block : kPairTickBlock;
    const int D = B > 1 ? std::min(ticks, block) : 1;
    int dir = c->pair_dir & 1;
    std::vector<uint8_t> order;
    order.reserve((size_t)ticks * B);
    for (int t0 = 0; t0 < ticks; t0 += D) {
        for (int bi = 0; bi < B; ++bi)
            for (int j = 0; j < std::min(D, ticks - t0); ++j) order.push_back((uint8_t)(dir ? B - 1 - bi : bi));
        dir ^= 1;
    }
    const int q0 = resume ? ra.parity : 0;  // half of the scratch that holds pass 1 of the stream's first entry
    ra.armed = false;
    ow::TickGroupArgs ga;
    std::memset(&ga, 0, sizeof(ga));
    ga.pair_compact = 1;
    // the foam rates of this delta (wave_generator.gd:104-106): the same for every tick of the run
    for (int i = 0; i < count; ++i) {
        ow_cascade_params &p = params[i];
        p.foam_grow_rate = delta * p.foam_amount * 7.5;
        const double d = 10.0 - p.foam_amount;
        p.foam_decay_rate = delta * (d > 0.5 ? d : 0.5) * 1.15;
    }
    ow::FrameArgs args = run_frame_args(params, count);
    const int total = (int)order.size();
    {   // pass 1 of the stream's first entry: launched alone here, or by the previous run's last launch
        const int b = order[0];
        advance_slots(delta, params, count, first[b], sizes[b], ga.time1[0]);
        if (resume) {
            ++c->la.hits;
        } else {
            ga.slots2 = 0, ga.d2 = 0;
            ga.first1 = first[b], ga.slots1 = sizes[b], ga.d1 = 1;
            ga.tbase1[0] = (q0 & 1) * c->pair_slots;
            if (ow_status st = launch_merged(c, args, ga); st != OW_OK) return st;
        }
    }
    for (int i = 0; i < total; ++i) {
        const int b = order[i];
        ga.first2 = first[b];
        ga.slots2 = sizes[b];
        ga.tbase2[0] = ((q0 + i) & 1) * c->pair_slots;
        ga.slots1 = 0;
        bool arming = false;
        if (i + 1 < total) {
            const int nb = order[i + 1];
            advance_slots(delta, params, count, first[nb], sizes[nb], ga.time1[0]);  // that batch's next tick
            ga.first1 = first[nb];
            ga.slots1 = sizes[nb];
        } else if (speculate) {  // the run's last launch: the next tick of the batch the run ends on -- what a run like this one would start with (dir has turned)
            for (int s = first[b]; s < first[b] + sizes[b]; ++s) {
                const ow_cascade_params &p = params[count - 1 - s];
                ga.time1[0][s] = ra.time[0][s] = (float)(p.time + delta);
            }
            for (int s = 0; s < count; ++s) {
                ra.tile_x[s] = params[count - 1 - s].tile_length[0];
                ra.tile_y[s] = params[count - 1 - s].tile_length[1];
            }
            ga.first1 = first[b];
            ga.slots1 = sizes[b];
            ra.kind = 2, ra.count = count, ra.batch = b, ra.first = first[b], ra.size = sizes[b], ra.parity = (q0 + i + 1) & 1;
            arming = true;
        }
        ga.tbase1[0] = ((q0 + i + 1) & 1) * c->pair_slots;
        ga.d2 = 1;
        ga.d1 = ga.slots1 > 0;
        if (ow_status st = launch_merged(c, args, ga); st != OW_OK) return st;
        if (arming) {
            ra.armed = true;
            ++c->la.speculated;
        }
    }
    c->pair_dir = dir;
    finish_merged_run(c, args, params, count, sizes[order[total - 1]], 6, 1);
    return OW_OK;
}

ow_status run_impl(ow_context *c, double delta, ow_cascade_params *params, int32_t count, int32_t frames) {
    // (measurements only: an irregular cadence for the call-by-call forms below)
    auto delta_of = [&](int tick) { return c->run_delta_period > 0 && ((tick / c->run_delta_period) & 1) ? delta * 1.25 : delta; };
    if (c->run_as_reference) {  // measurement: the reference's own schedule, call by call
        for (int f = 0; f < frames; ++f) {
            if (ow_status st = ow_update(c, delta_of(f), params, count); st != OW_OK) return st;
            for (int k = 0; k < count; ++k)
                if (ow_status st = ow_process(c); st != OW_OK) return st;
        }
        return OW_OK;
    }
    int f = 0;
    // SINGLE-BATCH ticks of the compact family (1024^2 x 2 .. 4, 512^2 x 7 .. 8, 2048^2 x 1): the run is a tick-by-tick caller of ow_update_all whose
    // next ticks are CERTAIN -- every tick launches its pass 2 together with pass 1 of the next one (k_tick_pair_c), the last tick of the run with
    // a speculated pass 1 of the tick a NEXT run would start with (the cadence rule of ow_update_all: after a run of equal deltas, one more).
    // Back-to-back runs are then one seamless stream of full pair launches: no ordinary first tick, no half-filled launches at the two ends of
    // every run -- round 5 found the driver's 20-tick regions paying 2 us per tick for those (profiles/r05_run_overhead.txt).  Every tick is
    // checked like any ow_update_all (records, dirty spectra, leftovers: the ordinary path takes what does not qualify); results bit-identical.
    if (frames >= 3 && !c->run_as_calls && !c->no_merge && c->timing == 0 && std::isfinite(delta) && count >= 1 && count <= c->cascades &&
        lookahead_mode(c, count) == 1) {
        const uint64_t before = c->la.hits + c->la.speculated;
        ow_status st = OW_OK;
        for (; f < frames && st == OW_OK; ++f) {
            c->la.certain = frames - f - 1;
            st = ow_update_all(c, delta, params, count);
        }
        c->la.certain = 0;
        if (st != OW_OK) return st;
        if (c->la.hits + c->la.speculated != before) {  // (the pair kernel did launch: what ow_last_kernel_family / ow_tick_group_depth report for a run)
            c->last_family = 6;
            c->last_group_depth = 1;
        }
        return OW_OK;
    }
    const bool may_merge = !c->run_as_calls && !c->no_merge && std::isfinite(delta) && count >= 1 && count <= c->cascades;
    // a run that follows a run (same delta, same cascades, nothing in between) lets its last launch work ahead for the next one
    const bool speculate = may_merge && c->ra.run_streak >= 1 && c->timing == 0 && !c->inject_fault;
    // ... and the next one starts in the middle of the stream: no ordinary first tick, no launch of pass 1 alone
    if (may_merge && frames >= 1 && c->ra.armed) {
        const int depth = run_resume_usable(c, delta, params, count, frames);
        if (depth != 0) {
            OW_HIP(hipSetDevice(c->device));
            for (int i = 0; i < count; ++i)
                if (params[i].should_generate_spectrum) {  // (resident: checked)
                    params[i].should_generate_spectrum = 0;
                    ++c->spectra_skipped;
                }
            note_cadence(c, delta);  // (what the first tick's ow_update_all would have noted)
            return depth > 0 ? run_tick_groups(c, delta, params, count, frames, depth, true, speculate) : run_tick_pairs(c, delta, params, count, frames, true, speculate);
        }
    }
    // the first tick takes the ordinary path (flush of leftovers, spectrum generation, validation of the records) ...
    if (frames >= 1) {
        c->la.hold = frames >= 3 && may_merge;  // (the run's own merged launches take over from tick 2 on)
        ow_status st = ow_update_all(c, delta, params, count);
        c->la.hold = false;
        if (st != OW_OK) return st;
        f = 1;
    }
    // ... the rest of a small batch goes out as tick groups (results identical: same lane code, same order per texel)
    int depth = frames - f >= 2 && may_merge ? tick_groups_usable(c, params, count) : 0;
    if (depth != 0) {
        OW_HIP(hipSetDevice(c->device));
        // the merged launches keep several ticks of intermediate in flight; if that scratch cannot be had the run is not lost: it goes
        // out one launch per pass (same results)
        if (ensure_scratch(c, depth > 0 ? 2 * depth * count : 2 * c->pair_slots) != OW_OK) depth = 0;
    }
    if (depth != 0) {
        ow_status st = depth > 0 ? run_tick_groups(c, delta, params, count, frames - f, depth, false, speculate) : run_tick_pairs(c, delta, params, count, frames - f, false, speculate);
        if (st != OW_OK) return st;
        f = frames;
    }
    for (; f < frames; ++f) {
        ow_status st = ow_update_all(c, c->run_as_calls ? delta_of(f) : delta, params, count);
        if (st != OW_OK) return st;
    }
    return OW_OK;
}

}  // namespace

ow_status ow_run(ow_context *c, double delta, ow_cascade_params *params, int32_t count, int32_t frames) {
    if (frames < 0) return fail(OW_ERR_INVALID, "frames must be >= 0");
    if (!c || !params) return fail(OW_ERR_INVALID, "null argument");
    // does this run follow a run like itself, with nothing in between?  (update_impl lowers last_was_run when it is called from outside a run)
    ow_context::RunAhead &ra = c->ra;
    const bool follows = ra.last_was_run && ra.last_count == count && std::memcmp(&ra.last_delta, &delta, sizeof(double)) == 0;
    ra.run_streak = follows ? std::min(ra.run_streak + 1, 1 << 20) : 0;
    c->inside_run = true;
    c->run_frames = frames;
    const ow_status st = run_impl(c, delta, params, count, frames);
    c->inside_run = false;
    c->run_frames = 0;
    ra.last_was_run = st == OW_OK && frames >= 1;
    ra.last_count = count;
    ra.last_delta = delta;
    if (st != OW_OK) ra.armed = false;
    join_for_caller(c);
    return st;
}

int32_t ow_last_kernel_family(const ow_context *c) { return c ? c->last_family : 0; }
int32_t ow_last_batch_cascades(const ow_context *c) { return c ? c->last_count : 0; }
int32_t ow_tick_group_depth(const ow_context *c) {
    if (!c) return 0;
    return c->last_family >= 5 ? c->last_group_depth : c->group_depth;
}

int32_t ow_cascades_remaining(const ow_context *c) { return c ? c->pass_num_cascades_remaining : 0; }

ow_status ow_sync(ow_context *c) {
    if (!c) return fail(OW_ERR_INVALID, "null context");
    OW_HIP(hipSetDevice(c->device));
    return sync_stream(c, 0u);
}

ow_status ow_get_device_ptrs(ow_context *c, void **disp, void **norm, size_t *stride) {
    if (!c) return fail(OW_ERR_INVALID, "null context");
    if (disp) *disp = c->buf.disp;
    if (norm) *norm = c->buf.norm;
    if (stride) *stride = plane(c) * sizeof(ow::u16x4);
    return OW_OK;
}

// ---- zero-copy hand-off: dma-buf export of the two arrays, import of a foreign allocation (ocean_waves.h) ----
ow_status ow_export_maps(ow_context *c, int32_t *disp_fd, int32_t *norm_fd, size_t *bytes_each) {
    if (!c) return fail(OW_ERR_INVALID, "null context");
    OW_HIP(hipSetDevice(c->device));
    const size_t bytes = (size_t)c->layers * plane(c) * sizeof(ow::u16x4);
    int fds[2] = {-1, -1};
    void *ptrs[2] = {c->buf.disp, c->buf.norm};
    const bool own[2] = {c->own_disp, c->own_norm};
    for (int i = 0; i < 2; ++i) {
        if ((i == 0 && !disp_fd) || (i == 1 && !norm_fd)) continue;
        if (!own[i]) {  // caller-owned memory: exportable only if it is a whole buffer object (see exportable_bytes)
            void *base = nullptr;
            size_t range = 0;
            if (hipMemGetAddressRange((hipDeviceptr_t *)&base, &range, (hipDeviceptr_t)ptrs[i]) != hipSuccess || base != ptrs[i] || range < kExportGranule ||
                ((uintptr_t)base & (kExportGranule - 1)) != 0) {
                (void)hipGetLastError();
                if (fds[0] >= 0) close(fds[0]);
                return fail(OW_ERR_STATE, "the caller-owned %s array is not the start of an allocation of at least 2 MiB: a dma-buf covers whole buffer "
                                          "objects, so the descriptor would not map the array from offset 0", i ? "normal" : "displacement");
            }
        }
        const hipError_t e = hipMemGetHandleForAddressRange(&fds[i], (hipDeviceptr_t)ptrs[i], bytes, hipMemRangeHandleTypeDmaBufFd, 0);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            if (fds[0] >= 0) close(fds[0]);
            return fail(OW_ERR_HIP, "hipMemGetHandleForAddressRange (dma-buf export of the %s array, %zu bytes) failed: %s", i ? "normal" : "displacement",
                        bytes, hipGetErrorString(e));
        }
    }
    if (disp_fd) *disp_fd = fds[0];
    if (norm_fd) *norm_fd = fds[1];
    if (bytes_each) *bytes_each = bytes;
    return OW_OK;
}

struct ow_imported {
    hipExternalMemory_t ext = nullptr;
    void *ptr = nullptr;
    int device = 0;
    int fd = -1;  // our duplicate of the caller's descriptor: this runtime neither takes ownership of an imported descriptor nor closes it
                  // (measured: forty import / release cycles left forty descriptors open), so it is closed by ow_release_buffer
};

ow_status ow_import_buffer(int32_t device_id, int32_t fd, size_t offset, size_t bytes, ow_imported **out, void **device_ptr) {
    if (!out || !device_ptr) return fail(OW_ERR_INVALID, "null argument");
    *out = nullptr;
    *device_ptr = nullptr;
    if (fd < 0 || bytes == 0) return fail(OW_ERR_INVALID, "bad file descriptor or size");
    int ndev = 0, caller_dev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(OW_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    OW_HIP(hipGetDevice(&caller_dev));
    const int dev = device_id < 0 ? caller_dev : device_id;
    if (dev >= ndev) return fail(OW_ERR_INVALID, "device_id %d >= device count %d", dev, ndev);
    struct DeviceRestore {
        int dev;
        ~DeviceRestore() { (void)hipSetDevice(dev); }
    } restore{caller_dev};
    OW_HIP(hipSetDevice(dev));
    const int own = dup(fd);  // the import consumes a descriptor; the caller's stays the caller's
    if (own < 0) return fail(OW_ERR_INVALID, "dup(%d) failed", fd);
    hipExternalMemoryHandleDesc hd;
    std::memset(&hd, 0, sizeof(hd));
    hd.type = hipExternalMemoryHandleTypeOpaqueFd;
    hd.handle.fd = own;
    hd.size = offset + bytes;
    hipExternalMemory_t ext = nullptr;
    hipError_t e = hipImportExternalMemory(&ext, &hd);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        close(own);
        return fail(OW_ERR_HIP, "hipImportExternalMemory(fd, %zu bytes) failed: %s", offset + bytes, hipGetErrorString(e));
    }
    hipExternalMemoryBufferDesc bd;
    std::memset(&bd, 0, sizeof(bd));
    bd.offset = 0;             // the window is applied to the mapped pointer below: given a non-zero offset, this runtime's
    bd.size = offset + bytes;  // GetMappedBuffer did not return the window's bytes (round 3, first version of tests/test_interop.py)
    void *ptr = nullptr;
    e = hipExternalMemoryGetMappedBuffer(&ptr, ext, &bd);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipDestroyExternalMemory(ext);
        close(own);
        return fail(OW_ERR_HIP, "hipExternalMemoryGetMappedBuffer failed: %s", hipGetErrorString(e));
    }
    ow_imported *im = new (std::nothrow) ow_imported();
    if (!im) {
        (void)hipDestroyExternalMemory(ext);
        close(own);
        return fail(OW_ERR_NOMEM, "out of host memory");
    }
    im->fd = own;
    im->ext = ext;
    im->ptr = ptr;
    im->device = dev;
    *out = im;
    *device_ptr = static_cast<char *>(ptr) + offset;
    return OW_OK;
}

void ow_release_buffer(ow_imported *im) {
    if (!im) return;
    int caller_dev = -1;
    (void)hipGetDevice(&caller_dev);
    (void)hipSetDevice(im->device);
    (void)hipDestroyExternalMemory(im->ext);
    if (im->fd >= 0) close(im->fd);
    delete im;
    if (caller_dev >= 0) (void)hipSetDevice(caller_dev);
}

ow_status ow_get_maps(ow_context *c, int32_t cascade, void *disp, void *norm) {
    ow_status st = check_cascade(c, cascade);
    if (st != OW_OK) return st;
    OW_HIP(hipSetDevice(c->device));
    const size_t bytes = plane(c) * sizeof(ow::u16x4);
    if (disp) OW_HIP(hipMemcpyAsync(disp, c->buf.disp + cascade * plane(c), bytes, hipMemcpyDeviceToHost, main_stream(c)));
    if (norm) OW_HIP(hipMemcpyAsync(norm, c->buf.norm + cascade * plane(c), bytes, hipMemcpyDeviceToHost, main_stream(c)));
    return sync_stream(c, 1u << cascade);
}

ow_status ow_readback_begin(ow_context *c, uint32_t mask) {
    if (!c) return fail(OW_ERR_INVALID, "null context");
    if (mask == 0 || (mask >> c->layers) != 0) return fail(OW_ERR_INVALID, "cascade_mask 0x%x selects no layer or one >= %d", mask, c->layers);
    OW_HIP(hipSetDevice(c->device));
    const size_t pl = plane(c), L = (size_t)c->layers, bytes = pl * sizeof(ow::u16x4);
    if (!c->snap_host) {  // first use: second stream, snapshot planes, page-locked staging, events -- committed only when ALL succeeded
        hipStream_t cs = nullptr;
        ow::u16x4 *sd = nullptr, *sh = nullptr;
        hipEvent_t ready[OW_MAX_CASCADES] = {}, done[OW_MAX_CASCADES] = {};
        bool ok = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) == hipSuccess && hipMalloc((void **)&sd, 2 * L * bytes) == hipSuccess &&
                  hipHostMalloc((void **)&sh, 2 * L * bytes, hipHostMallocDefault) == hipSuccess;
        for (size_t i = 0; ok && i < L; ++i)
            ok = hipEventCreateWithFlags(&ready[i], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&done[i], hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            const hipError_t e = hipGetLastError();
            for (auto &x : ready)
                if (x) (void)hipEventDestroy(x);
            for (auto &x : done)
                if (x) (void)hipEventDestroy(x);
            if (sh) (void)hipHostFree(sh);
            (void)hipFree(sd);
            if (cs) (void)hipStreamDestroy(cs);
            return fail(OW_ERR_NOMEM, "readback resources (2 x %zu bytes, device + page-locked host) could not be created: %s", 2 * L * bytes, hipGetErrorString(e));
        }
        c->copy_stream = cs;
        c->snap_dev = sd;
        c->snap_host = sh;
        for (size_t i = 0; i < L; ++i) {
            c->snap_ready[i] = ready[i];
            c->copy_done[i] = done[i];
        }
    }
    for (int i = 0; i < c->layers; ++i) {
        if (!((mask >> i) & 1u)) continue;
        ow::u16x4 *sd = c->snap_dev + (size_t)i * pl, *sn = c->snap_dev + (L + i) * pl;
        ow::u16x4 *hd = c->snap_host + (size_t)i * pl, *hn = c->snap_host + (L + i) * pl;
        // the snapshot slot may still be feeding an earlier PCIe copy
        if (c->copy_pending[i]) OW_HIP(hipStreamWaitEvent(main_stream(c), c->copy_done[i], 0));
        c->readback_faulted &= ~(1u << i);  // a new snapshot of this layer
        OW_HIP(hipMemcpyAsync(sd, c->buf.disp + (size_t)i * pl, bytes, hipMemcpyDeviceToDevice, main_stream(c)));
        OW_HIP(hipMemcpyAsync(sn, c->buf.norm + (size_t)i * pl, bytes, hipMemcpyDeviceToDevice, main_stream(c)));
        OW_HIP(hipEventRecord(c->snap_ready[i], main_stream(c)));
        OW_HIP(hipStreamWaitEvent(c->copy_stream, c->snap_ready[i], 0));
        OW_HIP(hipMemcpyAsync(hd, sd, bytes, hipMemcpyDeviceToHost, c->copy_stream));
        OW_HIP(hipMemcpyAsync(hn, sn, bytes, hipMemcpyDeviceToHost, c->copy_stream));
        OW_HIP(hipEventRecord(c->copy_done[i], c->copy_stream));
        c->copy_pending[i] = true;
    }
    return OW_OK;
}

ow_status ow_readback_wait(ow_context *c, int32_t cascade, const void **disp, const void **norm) {
    ow_status st = check_cascade(c, cascade);
    if (st != OW_OK) return st;
    if (!c->copy_stream || !c->copy_pending[cascade]) return fail(OW_ERR_STATE, "no readback of cascade %d is outstanding", cascade);
    OW_HIP(hipSetDevice(c->device));
    OW_HIP(hipEventSynchronize(c->copy_done[cascade]));
    c->copy_pending[cascade] = false;
    if (*c->status_host != 0u) {  // a frame kernel reported a failure: the bytes that landed are not maps
        c->readback_faulted |= 1u << cascade;  // (copy_pending was just cleared: mark this layer by hand, the others are marked by sync_stream)
        (void)sync_stream(c, 0u);
    }
    if ((c->readback_faulted >> cascade) & 1u) {  // every layer of the faulted batch fails, not only the first one waited for
        c->readback_faulted &= ~(1u << cascade);
        return fail(OW_ERR_HIP, "the readback of cascade %d carries maps of a batch that reported a device-side failure", cascade);
    }
    const size_t pl = plane(c), L = (size_t)c->layers;
    if (disp) *disp = c->snap_host + (size_t)cascade * pl;
    if (norm) *norm = c->snap_host + (L + cascade) * pl;
    return OW_OK;
}

ow_status ow_sample_surface(ow_context *c, const float *xz, int32_t count, const float *map_scales, int32_t num_cascades,
                            ow_surface_sample *out) {
    static_assert(sizeof(ow_surface_sample) == sizeof(ow::SurfaceSample) && sizeof(ow_surface_sample) == 64, "record layout");
    if (!c) return fail(OW_ERR_INVALID, "null context");
    if (count < 0) return fail(OW_ERR_INVALID, "count must be >= 0");
    if (num_cascades < 1 || num_cascades > c->cascades) return fail(OW_ERR_INVALID, "num_cascades %d outside [1,%d]", num_cascades, c->cascades);
    if (count == 0) return OW_OK;
    if (!xz || !map_scales || !out) return fail(OW_ERR_INVALID, "null argument");
    OW_HIP(hipSetDevice(c->device));
    if (count > c->query_capacity) {
        (void)hipFree(c->query_xz);
        (void)hipFree(c->query_out);
        c->query_xz = nullptr;
        c->query_out = nullptr;
        c->query_capacity = 0;
        const int cap = std::max(count, 4096);
        if (hipMalloc((void **)&c->query_xz, (size_t)cap * 2 * sizeof(float)) != hipSuccess ||
            hipMalloc((void **)&c->query_out, (size_t)cap * sizeof(ow::SurfaceSample)) != hipSuccess)
            return fail(OW_ERR_NOMEM, "hipMalloc failed for %d query points", cap);
        c->query_capacity = cap;
    }
    ow::SurfaceScales sc;
    std::memset(&sc, 0, sizeof(sc));
    std::memcpy(sc.s, map_scales, (size_t)num_cascades * 4 * sizeof(float));
    OW_HIP(hipMemcpyAsync(c->query_xz, xz, (size_t)count * 2 * sizeof(float), hipMemcpyHostToDevice, main_stream(c)));
    OW_HIP(ow::launch_sample_surface(c->n, num_cascades, c->buf, c->query_xz, count, sc, c->query_out, main_stream(c)));
    OW_HIP(hipMemcpyAsync(out, c->query_out, (size_t)count * sizeof(ow::SurfaceSample), hipMemcpyDeviceToHost, main_stream(c)));
    return sync_stream(c, (1u << num_cascades) - 1u);
}

ow_status ow_set_normal_map(ow_context *c, int32_t cascade, const void *norm) {
    ow_status st = check_cascade(c, cascade);
    if (st != OW_OK) return st;
    if (!norm) return fail(OW_ERR_INVALID, "null normal map");
    OW_HIP(hipSetDevice(c->device));
    // the foam channel (.a) also goes into the context's private FP16 foam plane, in pass-2 lane order
    // (ow_device.h Pass2::foam_index): that plane, not normal.a, is what the recurrence reads back
    const size_t n = (size_t)c->n, tl = n / 16;
    std::vector<uint16_t> foam(plane(c));
    const uint16_t *src = static_cast<const uint16_t *>(norm);
    for (size_t xp = 0; xp < n; ++xp)
        for (size_t yp = 0; yp < n; ++yp) foam[xp * n + (yp % tl) * 16 + yp / tl] = src[(xp * n + yp) * 4 + 3];
    OW_HIP(hipMemcpyAsync(c->buf.norm + cascade * plane(c), norm, plane(c) * sizeof(ow::u16x4), hipMemcpyHostToDevice, main_stream(c)));
    OW_HIP(hipMemcpyAsync(c->buf.foam + cascade * plane(c), foam.data(), plane(c) * sizeof(uint16_t), hipMemcpyHostToDevice, main_stream(c)));
    OW_HIP(hipStreamSynchronize(main_stream(c)));
    return OW_OK;
}

ow_status ow_get_maps_f32(ow_context *c, int32_t cascade, float *out) {
    ow_status st = check_cascade(c, cascade);
    if (st != OW_OK) return st;
    if (!c->buf.f32) return fail(OW_ERR_STATE, "context was created without OW_FLAG_DEBUG_F32");
    if (!out) return fail(OW_ERR_INVALID, "null output");
    OW_HIP(hipSetDevice(c->device));
    OW_HIP(hipMemcpyAsync(out, c->buf.f32 + cascade * plane(c) * 8, plane(c) * 8 * sizeof(float), hipMemcpyDeviceToHost, main_stream(c)));
    return sync_stream(c, 1u << cascade);
}

ow_status ow_get_spectrum(ow_context *c, int32_t cascade, float *h0, float *omega) {
    ow_status st = check_cascade(c, cascade);
    if (st != OW_OK) return st;
    OW_HIP(hipSetDevice(c->device));
    std::vector<ow::cplx> a;
    if (h0) {
        a.resize(plane(c));
        OW_HIP(hipMemcpyAsync(a.data(), c->buf.h0 + cascade * plane(c), plane(c) * sizeof(ow::cplx), hipMemcpyDeviceToHost, main_stream(c)));
    }
    if (omega) OW_HIP(hipMemcpyAsync(omega, c->buf.omega + cascade * plane(c), plane(c) * sizeof(float), hipMemcpyDeviceToHost, main_stream(c)));
    OW_HIP(hipStreamSynchronize(main_stream(c)));
    if (h0) {  // rebuild the reference's texel (h0(k), conj(h0(-k))) from the stored half (spectrum_compute.glsl:121-124)
        const size_t n = (size_t)c->n;
        for (size_t y = 0; y < n; ++y)
            for (size_t x = 0; x < n; ++x) {
                const ow::cplx v = a[y * n + x], m = a[((n - y) % n) * n + (n - x) % n];
                float *o = h0 + (y * n + x) * 4;
                o[0] = v.x;
                o[1] = v.y;
                o[2] = m.x;
                o[3] = -m.y;
            }
    }
    return OW_OK;
}

ow_status ow_get_push_constants(const ow_context *c, int32_t cascade, ow_push_constants *out) {
    ow_status st = check_cascade(c, cascade);
    if (st != OW_OK) return st;
    if (!out) return fail(OW_ERR_INVALID, "null output");
    if (cascade >= OW_MAX_CASCADES || !c->pc_valid[cascade]) return fail(OW_ERR_STATE, "cascade %d has not been launched yet", cascade);
    *out = c->pc_words[cascade];
    return OW_OK;
}

ow_status ow_get_intermediate(ow_context *c, int32_t cascade, float *out) {
    ow_status st = check_cascade(c, cascade);
    if (st != OW_OK) return st;
    if (!out) return fail(OW_ERR_INVALID, "null output");
    OW_HIP(hipSetDevice(c->device));
    const size_t pl = plane(c), n = (size_t)c->n;
    std::vector<ow::cplx> t(pl * ow::kLayers);
    if (c->slot_of[cascade] < 0)
        return fail(OW_ERR_STATE, "cascade %d was not part of the most recent batch: its intermediate has been overwritten", cascade);
    if (c->last_family >= 3)
        return fail(OW_ERR_STATE, "the most recent batch used the compact (three-layer) intermediate, which has no counterpart in the "
                                  "reference's fft_buffer: create the context with OW_FLAG_KERNELS_STANDARD to inspect it");
    OW_HIP(hipMemcpyAsync(t.data(), c->buf.T + (size_t)c->slot_of[cascade] * pl * ow::kLayers, t.size() * sizeof(ow::cplx), hipMemcpyDeviceToHost, main_stream(c)));
    OW_HIP(hipStreamSynchronize(main_stream(c)));
    // device layout T[layer][y/16][x'][y%16]  ->  reference half-0-after-transpose layout [layer][row = x'][col = y].
    // The device rows carry the x' half of the ifftshift sign, (-1)^x' (see Pass1::rot): taken out again here.
    for (int layer = 0; layer < ow::kLayers; ++layer)
        for (size_t xp = 0; xp < n; ++xp) {
            const float sgn = (xp & 1) ? -1.0f : 1.0f;
            for (size_t y = 0; y < n; ++y) {
                const ow::cplx v = t[ow::t_unit((int)n, layer, (int)xp, (int)y)];
                float *o = out + (((size_t)layer * n + xp) * n + y) * 2;
                o[0] = v.x * sgn;
                o[1] = v.y * sgn;
            }
        }
    return OW_OK;
}

ow_status ow_probe_kernel_times(ow_context *c, int32_t reps, float *p1_ms, float *p2_ms, int32_t *cascades_per_launch) {
    if (!c) return fail(OW_ERR_INVALID, "null context");
    if (reps < 1 || c->last_count < 1) return fail(OW_ERR_STATE, "nothing has been launched yet (or reps < 1)");
    OW_HIP(hipSetDevice(c->device));
    c->la.armed = false;  // (the probe launches write the scratch intermediate from slot 0 on)
    c->ra.armed = false;
    hipEvent_t e[3] = {};
    float a = 0, b = 0;
    auto run = [&]() -> ow_status {
        for (auto &x : e) OW_HIP(hipEventCreate(&x));
        OW_HIP(hipEventRecord(e[0], main_stream(c)));
        for (int i = 0; i < reps; ++i) OW_HIP(ow::launch_pass1(c->n, c->last_count, c->kernel_mode, c->last_args, c->buf, main_stream(c)));
        OW_HIP(hipEventRecord(e[1], main_stream(c)));
        for (int i = 0; i < reps; ++i) OW_HIP(ow::launch_pass2(c->n, c->last_count, c->kernel_mode, c->last_args, c->buf, main_stream(c)));
        OW_HIP(hipEventRecord(e[2], main_stream(c)));
        OW_HIP(hipEventSynchronize(e[2]));
        OW_HIP(hipEventElapsedTime(&a, e[0], e[1]));
        OW_HIP(hipEventElapsedTime(&b, e[1], e[2]));
        return OW_OK;
    };
    const ow_status st = run();
    for (auto &x : e)
        if (x) (void)hipEventDestroy(x);  // on every way out
    if (st != OW_OK) return st;
    if (p1_ms) *p1_ms = a / reps;
    if (p2_ms) *p2_ms = b / reps;
    if (cascades_per_launch) *cascades_per_launch = c->last_count;
    return OW_OK;
}

ow_status ow_timing_enable(ow_context *c, int32_t enable) {
    if (!c) return fail(OW_ERR_INVALID, "null context");
    if (!enable) {
        ow_status st = collect_timing(c);
        if (st != OW_OK) return st;
    }
    c->timing = enable == 2 ? 2 : enable != 0;
    return OW_OK;
}

ow_status ow_timing_read(ow_context *c, float *p1, float *p2, int32_t *launches, int32_t reset) {
    if (!c) return fail(OW_ERR_INVALID, "null context");
    ow_status st = collect_timing(c);
    if (st != OW_OK) return st;
    const int n = c->t_launches;
    if (p1) *p1 = n ? (float)(c->t1_ms / n) : 0.0f;
    if (p2) *p2 = n ? (float)(c->t2_ms / n) : 0.0f;
    if (launches) *launches = n;
    if (reset) {
        c->t1_ms = c->t2_ms = 0;
        c->t_launches = 0;
    }
    return OW_OK;
}

ow_status ow_timing_read_launches(ow_context *c, float *launch_ms_avg, int32_t *launches, int32_t reset) {
    if (!c) return fail(OW_ERR_INVALID, "null context");
    ow_status st = collect_timing(c);
    if (st != OW_OK) return st;
    if (launch_ms_avg) *launch_ms_avg = c->tg_launches ? (float)(c->tg_ms / c->tg_launches) : 0.0f;
    if (launches) *launches = c->tg_launches;
    if (reset) {
        c->tg_ms = 0;
        c->tg_launches = 0;
    }
    return OW_OK;
}

}  // extern "C"
