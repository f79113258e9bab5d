/*
 * ocean_waves.h -- C-ABI of the MI355X-native ocean-wave generator (libocean_waves.so).
 *
 * Drop-in boundary for ONE path of 2Retr0/GodotOceanWaves: the per-cascade
 *   spectrum -> time-modulate -> 2-D inverse FFT -> unpack / Jacobian / foam
 * pipeline that `WaveGenerator` (assets/water/wave_generator.gd) drives through six GLSL compute
 * shaders (assets/shaders/compute/).  Each entry point names the reference interface it replaces
 * (file:line, paths relative to the reference checkout).  Plain C: POD structs, pointers and
 * sizes, int status codes, no callbacks, no exceptions across the boundary.  A context is not
 * thread-safe: one caller thread per context, exactly like the reference (everything runs on
 * Godot's main thread, wave_generator.gd:19).
 *
 * There is NO CPU fallback: ow_create() fails with OW_ERR_NO_DEVICE when no gfx950-class HIP
 * device is visible.
 */
#ifndef OCEAN_WAVES_H
#define OCEAN_WAVES_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OW_MAX_CASCADES 8 /* MAX_CASCADES, assets/shaders/spatial/water.gdshader:8 */
#define OW_MAX_DEVICES 8  /* the GPUs of one node */
#define OW_ABI_VERSION 4 /* 2: ow_update copies the records (no borrowed pointer), ow_set/get_cascade_params, device status word;
                            3: ow_group_* (cascades sharded over several devices, gather into the consumer's arrays), records are
                               validated on the way in (ow_update / ow_set_cascade_params), sticky device-side failures,
                               ow_export_maps / ow_import_buffer (dma-buf hand-off);
                            4: the scalar fields of ow_cascade_params are FP64, as a GDScript caller holds them, and are narrowed
                               where the reference narrows them (the push-constant pack) -- the record is 128 bytes */

typedef enum ow_status {
    OW_OK = 0,
    OW_ERR_INVALID = 1,   /* bad argument (reference: assert, wave_generator.gd:91, render_context.gd:77) */
    OW_ERR_NO_DEVICE = 2, /* no HIP device / wrong architecture */
    OW_ERR_HIP = 3,       /* a HIP runtime call failed, or a frame kernel reported a device-side failure through the
                             context's status word (found at the next synchronising call); see ow_last_error() */
    OW_ERR_NOMEM = 4,
    OW_ERR_STATE = 5      /* call order violated (e.g. ow_process with nothing armed is NOT an error: it is a no-op) */
} ow_status;

/* WaveCascadeParameters -- assets/water/wave_cascade_parameters.gd:7-42.
 * Field meaning, units and defaults are the reference's, and so are the TYPES a GDScript caller holds: every exported
 * `float` of the resource is FP64 there and is narrowed to FP32 only when it is packed into a push constant
 * (assets/render_context.gd:131-134, encode_float), AFTER the host math that uses it -- JONSWAP alpha / peak frequency
 * (wave_generator.gd:69-70,116-121), deg_to_rad (:71), foam_grow_rate / foam_decay_rate (:104-106).  So the scalar
 * parameters are `double` here (ABI 4) and the library narrows them exactly where the reference does; a caller that
 * holds FP32 values simply widens them.  `tile_length` is a Vector2, whose components are FP32 in Godot (real_t) and
 * stay `float`.  `time`, `foam_*_rate` and `should_generate_spectrum` are runtime state: ow_update advances them inside
 * the caller's struct (wave_generator.gd:103-106) during the call. */
typedef struct ow_cascade_params {
    float tile_length[2];      /* Vector2 (FP32 components): metres covered by the tile, default (50, 50)     :7  */
    double displacement_scale; /* consumer-side only, default 1.0                                             :9  */
    double normal_scale;       /* consumer-side only, default 1.0                                             :11 */
    double wind_speed;         /* m/s, default 20; values below 1e-4 are used as 1e-4 (the setter's clamp)    :15 */
    double wind_direction;     /* degrees, default 0                                                          :17 */
    double fetch_length;       /* km, default 550; values below 1e-4 are used as 1e-4                         :20 */
    double swell;              /* [0,2], default 0.8                                                          :22 */
    double spread;             /* [0,1], default 0.2                                                          :25 */
    double detail;             /* [0,1], default 1.0                                                          :28 */
    double whitecap;           /* [0,2], default 0.5                                                          :32 */
    double foam_amount;        /* [0,10], default 5.0                                                         :34 */
    int32_t spectrum_seed[2];  /* Vector2i, offsets the hash lattice                                          :37 */
    int32_t should_generate_spectrum; /* dirty flag, default 1                                              :38 */
    int32_t reserved;
    double time;               /* seconds; water.gd:32 starts cascade i at 120 + PI*i                         :40 */
    double foam_grow_rate;     /* set by ow_update: delta * foam_amount * 7.5                                 :41 */
    double foam_decay_rate;    /* set by ow_update: delta * max(0.5, 10 - foam_amount)*1.15                   :42 */
} ow_cascade_params;           /* 128 bytes */

/* Creation parameters -- replaces `wave_generator.map_size = N; wave_generator.init_gpu(C)`
 * (assets/water/water.gd:89-91, wave_generator.gd:8,17). */
typedef struct ow_config {
    int32_t map_size;     /* 128, 256, 512, 1024 (reference set, water.gd:38) or 2048 (beyond the reference) */
    int32_t num_cascades; /* 1..OW_MAX_CASCADES; like the reference, max(2, n) array layers are allocated (water.gd:91) */
    int32_t device_id;    /* HIP device ordinal; -1 = current device */
    float depth;          /* metres; the reference hard-codes DEPTH = 20.0 (wave_generator.gd:6); <= 0 selects 20 */
    void *stream;         /* hipStream_t to enqueue on; NULL = the context creates its own non-blocking stream */
    void *displacement_map; /* optional caller-owned DEVICE buffer, layers*N*N*8 bytes (RGBA16F); NULL = context allocates */
    void *normal_map;       /* optional caller-owned DEVICE buffer, same size.  Both buffers are ZEROED by ow_create (foam
                               starts from 0); the recurrence re-reads a private FP16 copy of .a, so a saved state is
                               restored with ow_set_normal_map, never by pre-loading this buffer */
    uint32_t flags;       /* OW_FLAG_* */
} ow_config;

#define OW_FLAG_DEBUG_F32 1u /* also keep 8 pre-quantisation FP32 channels per texel (parity tests) */
/* Kernel family.  By default the runtime picks per batch: the layer-parallel kernels (one lane group per row AND
 * packed layer) when a batch is too small to fill the chip; otherwise the compact-intermediate kernels where they
 * exist (map_size >= 256) and the standard ones (one lane group per row, four layers in sequence, the reference's
 * packing) elsewhere.  These flags pin the choice (tests, measurements). */
#define OW_FLAG_KERNELS_STANDARD 2u
#define OW_FLAG_KERNELS_LAYER_PARALLEL 4u
/* Compact-intermediate kernels (map_size >= 256; standard ones below that): two and a half packed layers cross the
 * intermediate instead of the reference's four; ow_get_intermediate is not available for batches that used them.
 * Together with OW_FLAG_KERNELS_LAYER_PARALLEL: the layer-parallel kernels on the compact intermediate (map_size >= 256). */
#define OW_FLAG_KERNELS_COMPACT 8u
/* ow_run merges launches across ticks where that pays (the first tick of a run takes the ordinary path unless the run continues a previous one, see ow_run):
 *  - TICK GROUPS, small batches (the layer-parallel compact family): one launch does pass 2 of up to four (256^2 x <= 4: eight)
 *    consecutive ticks (a block walks through the ticks of its rows, foam in registers) together with pass 1 of the next ones
 *    (independent of everything earlier) -- K / 4 + 1 launches for K ticks, and a chip that one small tick cannot fill is filled
 *    by several (k_tick_group_c_lp);
 *  - TICK PAIRS, the compact family (1024^2 x 2 .. 8, 512^2 x 7 .., 2048^2 x 1 .. 8): the run is a stream of batches of at most 4 Mi
 *    texels and one launch does pass 2 of one batch and pass 1 of the next -- the same cascades one tick later, or the tick's other
 *    cascades (k_tick_pair_c; k_tick_pair_c_split at 2048^2, where a batch is one cascade).  A tick of several batches runs each batch
 *    through a block of up to 64 ticks before the stream moves on to the next batch, so that a launch pairs a batch with itself one tick
 *    later and re-reads its spectra and foam from the Infinity Cache: cascades are independent, the state ow_run leaves behind is the same.
 * Results are bit-identical to one launch per pass; this flag keeps ow_run on one pair of launches per tick (tests, measurements).
 * The shipped library reads NOTHING from the environment.  (A/B builds compiled with -DOW_MEASUREMENT_KNOBS read the OW_DEBUG_* variables listed
 * in ow_runtime.hip plan_tick_groups once, in ow_create; one of them, OW_DEBUG_RUN_DELTA_CHANGE_EVERY, changes the deltas the call-by-call forms
 * of ow_run issue and with them the results.) */
#define OW_FLAG_NO_TICK_GROUPS 16u
/* ow_run issues its ticks exactly as an external caller of ow_update_all would, one call per tick (no merging across the ticks of the run);
 * ow_update_all's own adaptive look-ahead stays on.  For measuring what tick-by-tick callers get without a host round trip per tick. */
#define OW_FLAG_RUN_AS_CALLS 32u
/* ... and as the reference's own schedule: per tick one ow_update and `count` ow_process calls (wave_generator.gd:56-63,90-109). */
#define OW_FLAG_RUN_AS_REFERENCE_SCHEDULE 64u
/* Tests / measurements: pin the form of the tick groups' work items, which the runtime otherwise picks by batch size -- pass 1 as
 * layer-parallel items (one (8 rows, layer) per lane group) or as k_pass1c-shaped items (8 rows, all layers); pass 2 as plain blocks (a block
 * walks through the ticks of its columns) or pipelined ones (the block's two halves on alternate ticks).  Results do not depend on them. */
#define OW_FLAG_GROUP_P1_LP 0x100u
#define OW_FLAG_GROUP_P1_COMPACT 0x200u
#define OW_FLAG_GROUP_P2_PLAIN 0x400u
#define OW_FLAG_GROUP_P2_PIPE 0x800u
/* Tests / measurements: every raised should_generate_spectrum launches the spectrum kernel, as the reference's _update does
 * (wave_generator.gd:68-72), even when the record packs to the very constants the resident spectrum was generated from (ow_spectrum_stats). */
#define OW_FLAG_ALWAYS_REGENERATE_SPECTRUM 0x1000u
/* ow_create allocates the scratch intermediate of ONE batch only; what the look-ahead of ow_update_all / ow_process keeps in flight (the pair
 * kernel two batches, the group kernel a ring of five groups: 160 MiB at 1024^2 x 1, ~240 MiB at 512^2 x 8) is then allocated by the first call
 * that speculates (one hipMalloc + stream synchronisation inside that call, none afterwards).  For contexts that are only driven through ow_run's
 * tick groups, or many shards on one device; the default keeps every per-frame call free of allocations. */
#define OW_FLAG_LAZY_SCRATCH 0x2000u
/* TWO CHAINS (round 6; 1024^2 with four or more cascades, 512^2 with eight).  A tick-pair launch of four 1024^2 (eight 512^2) cascades on either side is two
 * generations of blocks (one), and the kernel boundary between two such launches costs a tenth of them (the chip drains and fills again).  Cascades are
 * independent: such a launch goes out as TWO launches of half the cascades each, the second on a stream of the context's own, each half a chain by itself -- one
 * chain's drain runs under the other's body (1024^2 x 4: 52.1 -> 48.0 us per tick on one box, 512^2 x 8: 27.3 -> 25.6; bit-identical maps: the same kernel on
 * the same items).  Everything else the context enqueues or waits for is ordered behind BOTH chains (ow_sync, the readbacks, ow_get_maps, ow_process, ... join
 * first).  Where the context runs on a stream of the CALLER's (ow_config.stream), the second chain is joined before the call returns, so that work the caller
 * enqueues on that stream afterwards finds every map complete, as before -- and because that join costs more than one split launch gains, on a caller's stream
 * only the launches of an ow_run of at least 8 ticks (16 at 512^2) are split (one join per run); ow_update_all tick by tick stays on the one stream there.  On the context's
 * own stream every such launch is split.  This flag keeps every launch whole, on the one stream (tests, A/B).  ow_chain_stats: launches that went out as two
 * chains. */
#define OW_FLAG_SINGLE_STREAM 0x4000u

typedef struct ow_context ow_context;

/* ---- lifetime ------------------------------------------------------------------------------ */

/* WaveGenerator.init_gpu (wave_generator.gd:17-54): allocates spectrum, FFT intermediate and the two
 * RGBA16F output arrays; uploads the twiddle tables (replaces the fft_butterfly dispatch, :52-54). */
ow_status ow_create(const ow_config *config, ow_context **out);

/* NOTIFICATION_PREDELETE -> context.free() (wave_generator.gd:111-113). NULL is allowed. */
void ow_destroy(ow_context *ctx);

/* Defaults of wave_cascade_parameters.gd:7-38. */
void ow_cascade_params_default(ow_cascade_params *p);

/* ---- the per-tick surface ---------------------------------------------------------------------- */

/* WaveGenerator.update(delta, parameters) (wave_generator.gd:90-109):
 *   1. cascades armed by the previous call and not yet processed are flushed now, indices
 *      0..remaining-1, with the PREVIOUS records (:94-98);
 *   2. for every cascade: time += delta, foam_grow_rate, foam_decay_rate (:101-106), written into `params`;
 *   3. all `count` cascades are armed (:108-109).
 * The reference keeps a reference to the caller's Array and reads the live objects later; a C caller's memory is
 * only borrowed DURING this call: the context keeps a COPY of the `count` records (a managed caller pins its array for
 * the call and no longer).  `should_generate_spectrum` is consumed: the armed copy carries it until the cascade is
 * processed, and it is cleared in `params`, so an unchanged array does not regenerate its spectra every tick.
 * Errors leave no trace: all `count` records are checked first (every field finite, tile_length positive, time + delta
 * finite) and a refused call (OW_ERR_INVALID) has advanced no time, consumed no dirty flag, armed and launched nothing -- the
 * corrected array simply goes in again.  Leftovers of the previous arm never survive this call: if their flush fails
 * (only a HIP failure can make it) they are dropped, `params` is still untouched, and the call can be repeated. */
ow_status ow_update(ow_context *ctx, double delta, ow_cascade_params *params, int32_t count);

/* "The parameter objects are live" made explicit: replace / read the context's copy of armed record `index`
 * (0 <= index < count of the last ow_update).  A caller that lets the user edit parameters between ow_update and the
 * ow_process that consumes them (the reference reads the edited object, wave_generator.gd:56-72) pushes the edited record
 * with ow_set_cascade_params before that ow_process; ow_get_cascade_params returns the record as the generator left it
 * (should_generate_spectrum cleared once the cascade has been processed, :72).  A record the kernels cannot take is refused
 * (OW_ERR_INVALID) and the armed copy stays as it was. */
ow_status ow_set_cascade_params(ow_context *ctx, int32_t index, const ow_cascade_params *params);
ow_status ow_get_cascade_params(const ow_context *ctx, int32_t index, ow_cascade_params *out);

/* WaveGenerator._process (wave_generator.gd:56-63): processes ONE armed cascade (highest index
 * first) -- the reference's one-cascade-per-rendered-frame load balancing.  No-op when nothing is armed.
 * The launch also carries pass 1 of the cascades the NEXT ow_process calls will take -- up to four of them (index - 1, index - 2, ..: their
 * armed records are known, nothing is guessed; behind an update's last cascade: the next update's cascades at time + delta, once the deltas
 * repeat), each checked when its call comes; the calls in between launch pass 2 alone.  A record edited in between (ow_set_cascade_params)
 * simply takes the ordinary two launches.  Bit-identical results.  (1024^2 x 4 on this schedule: 119 -> 85 us per update.)
 * Where nothing is waiting when an update arms its cascades -- the deltas of a scene behind water.gd's rate limiter never repeat -- ow_update
 * itself launches pass 1 of the cascades the ow_process calls will take (up to four, ONE launch that fills the chip; map sizes up to 1024),
 * and whatever an update leaves for the next one to flush (:94-98) is flushed from that queue instead of being recomputed. */
ow_status ow_process(ow_context *ctx);

/* Throughput mode: ow_update() followed by all armed cascades in ONE pair of kernel launches
 * (results identical to calling ow_process() `count` times).
 * Adaptive look-ahead: once two consecutive calls have come with the same delta, the call also launches a SPECULATED pass 1 of the next tick
 * (this tick's times + delta) together with its own pass 2; the next call checks the speculation against what it is actually given (count,
 * every FP32 time and tile length bit for bit, no spectrum to regenerate, nothing else has run in between) and, on a hit, costs one merged
 * launch instead of two.  Ticks of up to 1 Mi texels (the layer-parallel compact family) compute pass 1 of as many of the next ticks as the
 * caller's cadence predicts, up to FOUR, in one launch, and the calls in between launch pass 2 alone (1024^2 x 1: 29.9 -> 20.3 us per tick).  The
 * prediction: inside a run of equal deltas no further than the caller's previous run went, beyond it as far as this run has outlasted it (a
 * caller whose delta changes every k updates is never speculated across a change).  "The same delta" tolerates one nanosecond: a fixed-step scene
 * behind water.gd's rate limiter issues deltas that are equal up to the rounding noise of its FP64 clock, and what a hit needs is the FP32-narrowed
 * time, which is compared bit for bit anyway.  A miss discards the speculated work; results are bit-identical either way.  Single-batch ticks of
 * the compact families only (map_size >= 256; up to 4 Mi texels per tick); off under OW_FLAG_NO_TICK_GROUPS.  The scratch the look-ahead keeps in
 * flight (the pair kernel two batches, the group kernel a ring of five groups: at most a few hundred MiB) is allocated by ow_create: the per-frame
 * calls never allocate.  A device-side failure reported by a synchronising call also drops whatever had been computed ahead.
 * ow_lookahead_stats: calls served from work computed ahead, launches that carried some. */
ow_status ow_update_all(ow_context *ctx, double delta, ow_cascade_params *params, int32_t count);
ow_status ow_lookahead_stats(const ow_context *ctx, uint64_t *hits, uint64_t *speculated);

/* The dirty flag and the spectrum that is already there.  In the reference EVERY exported setter of WaveCascadeParameters raises
 * should_generate_spectrum -- `whitecap` and `foam_amount` included (wave_cascade_parameters.gd:32-35), which spectrum_compute.glsl never
 * reads -- and the next _update re-dispatches spectrum_compute with the SAME push constants (wave_generator.gd:68-72).  The spectrum is a
 * deterministic function of the thirteen packed words of that block (ow_get_push_constants: spectrum) and the map size, so a dirty record
 * that packs to exactly the words layer i's resident spectrum was generated from is served by what is there: the flag is consumed where the
 * record enters the context (ow_update / ow_update_all / ow_run, ow_set_cascade_params), no spectrum kernel is launched, and the record stays
 * on the merged launches and the look-ahead (which step aside for a spectrum that has to be generated).  Bit-identical maps; a whitecap
 * slider dragged at 50 updates per second no longer costs a spectrum per cascade per update.  ow_get_cascade_params then shows the flag
 * already cleared.  ow_spectrum_stats: spectrum kernels launched by this context, and dirty flags consumed without one. */
ow_status ow_spectrum_stats(const ow_context *ctx, uint64_t *generated, uint64_t *skipped);
ow_status ow_chain_stats(const ow_context *ctx, uint64_t *split_launches);   /* OW_FLAG_SINGLE_STREAM */

/* `frames` consecutive ow_update_all() ticks with the same delta, enqueued back to back (the reference's
 * "1000-frame loop" without a host round trip per tick).  Equivalent to calling ow_update_all `frames` times: only the state a run leaves
 * behind is defined (inside it the runtime may order independent cascades' ticks as it likes, see OW_FLAG_NO_TICK_GROUPS).
 * WORK LEFT IN THE QUEUE.  Runs that follow each other are one seamless stream of full launches, which means that the last launch of a run may
 * carry pass 1 of the tick(s) a NEXT run would start with, still in flight when ow_run returns (ow_sync / a readback wait for it like for anything
 * else; a next call that does not match discards it; the maps and every state a caller can observe are unaffected):
 *  - single-batch ticks of the compact family (1024^2 x 2 .. 4, 512^2 x 7 .. 8, 2048^2 x 1): the run's last tick speculates one more tick by
 *    ow_update_all's cadence rule (after a run of equal deltas: always);
 *  - tick groups and multi-batch tick pairs (256^2, 512^2 x <= 6, 1024^2 x 1; 1024^2 x 5 .. 8, 2048^2 x 2 .. 8): only a run that itself FOLLOWED a run
 *    with the same delta and cascade count, nothing in between, works ahead for the next one -- the first group of ticks, or the next tick of the
 *    batch the run ended on -- so a one-shot caller (one ow_run, then a readback) leaves nothing behind.  The next run checks it like a look-ahead
 *    hit (count, every FP32 time and tile length bit for bit, nothing armed, no spectrum to generate, nothing else has used the scratch) and then
 *    starts in the middle of the stream: no ordinary first tick, no half-filled launches at the ends of a run (ow_lookahead_stats counts both). */
ow_status ow_run(ow_context *ctx, double delta, ow_cascade_params *params, int32_t count, int32_t frames);

/* Number of armed, unprocessed cascades (pass_num_cascades_remaining, wave_generator.gd:15). */
int32_t ow_cascades_remaining(const ow_context *ctx);

/* Blocks until everything enqueued by this context has finished.  Also the point where a device-side failure shows: the
 * frame kernels OR a bit into the context's status word when a bounded wait gives up (the wave-pair rendezvous of the
 * 2048^2 kernels); a non-zero word turns this call -- and every other call that synchronises: ow_get_maps,
 * ow_get_maps_f32, ow_readback_wait, ow_sample_surface -- into OW_ERR_HIP.  The maps of the batches enqueued since the previous
 * synchronisation are then invalid, and so is the foam state they left behind (restore it with ow_set_normal_map).  The word
 * itself is consumed by the first call that sees it (ow_sync reports it once), but the failure is sticky for everything that
 * hands out map bytes, LAYER BY LAYER: ow_get_maps / ow_get_maps_f32 of a layer that one of those batches recomputed,
 * ow_sample_surface over such a layer, and the ow_readback_wait of EVERY layer whose copy was in flight keep returning OW_ERR_HIP
 * until a later batch has recomputed THAT layer (the reference's schedule enqueues one cascade per ow_process: the other layers
 * still hold the faulted batch's bytes) or, for a layer's readback, until its next ow_readback_begin.  The device-side wait is bounded by wall time (20 ms), and the report is a plain store + system fence
 * into page-locked host memory: it needs no PCIe atomics. */
ow_status ow_sync(ow_context *ctx);

/* ---- outputs: descriptors[&'displacement_map'/'normal_map'] (wave_generator.gd:34-35, water.gd:95-96) ---- */

/* Device pointers of the two RGBA16F array textures: layer-major [layer][row][col][4 x fp16],
 * layer stride = N*N*8 bytes.  Pixel (col = id.x, row = id.y) holds exactly what fft_unpack.glsl:50,67
 * imageStore()s at ivec3(id.x, id.y, cascade) -- including the transposed orientation that results
 * from skipping the second transpose (wave_generator.gd:77-82).  normal = (gradient.x, gradient.y,
 * dhx_dx, foam). */
ow_status ow_get_device_ptrs(ow_context *ctx, void **displacement_map, void **normal_map, size_t *layer_stride_bytes);

/* Host copy of one layer of each map in RenderingDevice.texture_update(tex, layer, bytes) layout
 * (row-major, 8 bytes per texel, N*N*8 bytes each).  Either pointer may be NULL.  Synchronises. */
ow_status ow_get_maps(ow_context *ctx, int32_t cascade, void *displacement_rgba16f, void *normal_rgba16f);

/* Foam / simulation state: the only persistent state besides `time` is the foam channel (normal.a, FP16,
 * fft_unpack.glsl:61-64).  The context keeps the bits the recurrence re-reads in a private FP16 plane (same
 * values as normal.a), so restoring state MUST go through ow_set_normal_map, which uploads N*N*8 bytes into one
 * layer and refreshes that plane (checkpoint/restore, re-sharding); writing into the normal map through the
 * device pointer does not change the simulation. */
ow_status ow_set_normal_map(ow_context *ctx, int32_t cascade, const void *normal_rgba16f);

/* ---- hand-off to a host-side consumer ------------------------------------------- */

/* Asynchronous readback of finished layers into page-locked host memory owned by the context: the bytes a
 * Godot-side shim passes to RenderingDevice.texture_update(tex, layer, bytes) (the maps are created with
 * TEXTURE_USAGE_CAN_UPDATE_BIT, wave_generator.gd:34-35 / render_context.gd:76-85).  `cascade_mask` bit i selects
 * layer i.  ow_readback_begin snapshots the selected layers in stream order (device-to-device, after everything
 * enqueued so far) and starts the PCIe copy on a second stream; it does not block, and later ow_update / ow_process
 * calls run concurrently with the copy.  ow_readback_wait blocks until the copy of one layer has landed and returns
 * pointers to N*N*8 bytes each (row-major RGBA16F); they stay valid until the next ow_readback_begin that selects
 * the same layer, or ow_destroy.  OW_ERR_STATE if no readback of that layer is outstanding. */
ow_status ow_readback_begin(ow_context *ctx, uint32_t cascade_mask);
ow_status ow_readback_wait(ow_context *ctx, int32_t cascade, const void **displacement_rgba16f, const void **normal_rgba16f);

/* ---- consumer-side sampling on the device ----------------------------------- */

/* What the reference's consumers evaluate at one world-space point (x, z) from the two array textures; texture() is
 * GL_LINEAR + GL_REPEAT with exact FP32 weights.  map_scales[i] = (1/tile_length.x, 1/tile_length.y,
 * displacement_scale, normal_scale) as built by water.gd:105-109. */
typedef struct ow_surface_sample {
    float displacement[3];    /* sum_i texture(displacements, vec3(xz*scales_i.xy, i)).xyz * scales_i.z
                                 (water.gdshader:31-37, sea_spray_particle.gdshader:103-108) */
    float gradient[2];        /* sum_i texture(normals, ...).xy, unscaled (sea_spray_particle.gdshader:80-82) */
    float gradient_scaled[2]; /* sum_i texture(normals, ...).xy * scales_i.w (water.gdshader:81, bilinear branch) */
    float foam;               /* sum_i texture(normals, ...).w */
    float normal_factor;      /* sea_spray_particle.gdshader:85: mix(0.25, 1, min((normal.y - 0.92) / 0.07, 1)) */
    float foam_factor;        /* :86: mix(0.25, 1, min((foam - 0.9) / 0.1, 1)) */
    float scale_factor;       /* :89 SCALE_FACTOR = normal_factor * foam_factor */
    int32_t spray_active;     /* :88 ACTIVE = normal_factor in [0,1] && foam > 0.9: the sea-spray spawn mask */
    float gradient_fragment[2]; /* water.gdshader:74-82 fragment(): sum_i mix(texture_bicubic, texture, min(1, 0.1 * map_size *
                                   min(scales_i.xy))).xy * scales_i.w -- the cubic B-spline filter of :41-68 included */
    float foam_fragment;        /* the same mix, .w */
    float reserved;
} ow_surface_sample;

/* Samples layers 0..num_cascades-1 at `count` points (world_xz = x0,z0,x1,z1,...; map_scales = 4 floats per cascade;
 * all host pointers) after everything enqueued so far, and writes `count` records.  Synchronises. */
ow_status ow_sample_surface(ow_context *ctx, const float *world_xz, int32_t count, const float *map_scales,
                            int32_t num_cascades, ow_surface_sample *out);

/* ---- several devices: cascades sharded inside one process ---------------------------------------- */

/* Cascades share nothing (wave_generator.gd:65-85 touches no state of another cascade; README.md:77-80), so a node's GPUs
 * each take a block of them: shard s owns the global cascades s*cascades_per_device .. +cascades_per_device-1 with all
 * their state (h0, foam, time) in its own ow_context on device_ids[s].  There is no data-path exchange.  The ONE
 * exchange is the gather of finished layers into the consumer's arrays on the root device -- the two RGBA16F array
 * textures water.gd:95-100 binds, layer g = global cascade g -- and only the owned layers travel:
 *     ow_group_gather_begin : per shard, in the shard's stream order: snapshot of the owned layers (device-to-device), then
 *                             on a side stream hipMemcpyPeerAsync over xGMI into the root's layer slots; returns at once,
 *                             later ticks overlap the transfer and may overwrite the live maps;
 *     ow_group_gather_wait  : blocks until every shard's layers have landed.
 * A shard on the root device itself copies straight into its slots (no second hop).  Each shard is driven by its own
 * worker thread (launches on N devices are enqueued side by side, not one device after the other); the group, like a
 * context, takes one caller thread. */
typedef struct ow_group_config {
    int32_t map_size;                   /* as ow_config */
    int32_t num_devices;                /* shards, 1..OW_MAX_DEVICES */
    int32_t device_ids[OW_MAX_DEVICES]; /* HIP ordinal of each shard; an ordinal may repeat (several shards on one device) */
    int32_t cascades_per_device;        /* >= 1; num_devices * cascades_per_device is the group's cascade count (beyond the reference's
                                           MAX_CASCADES = 8 the arrays are tiles of independent oceans, not one shader's cascades) */
    int32_t root;                       /* index into device_ids: the consumer's device, where the gathered arrays live */
    float depth;                        /* as ow_config */
    uint32_t flags;                     /* OW_FLAG_* for every shard, | OW_GROUP_FLAG_* */
    void *displacement_map;             /* optional caller-owned buffers ON THE ROOT DEVICE for the gathered arrays, */
    void *normal_map;                   /* max(2, cascades) * N * N * 8 bytes each; NULL = the group allocates */
} ow_group_config;
/* Test hook: treat every shard as remote (snapshot + side stream + hipMemcpyPeerAsync) even where it shares the root's device, so
 * that the whole peer path runs on a single-GPU box. */
#define OW_GROUP_FLAG_FORCE_PEER_PATH 0x10000u

typedef struct ow_group ow_group;

ow_status ow_group_create(const ow_group_config *config, ow_group **out);
void ow_group_destroy(ow_group *group);
int32_t ow_group_num_cascades(const ow_group *group);
/* the context of shard `shard` (borrowed: for per-shard queries such as ow_get_maps / ow_last_kernel_family; do not destroy) */
ow_context *ow_group_context(ow_group *group, int32_t shard);

/* ow_update / ow_process / ow_update_all / ow_run over the whole group: `params` holds the records of ALL cascades in global
 * order (count == ow_group_num_cascades), shard s works on its slice.  ow_group_process keeps the reference's order -- one armed
 * cascade per call, highest global index first (wave_generator.gd:56-63).  The first failing shard's status is returned; its
 * message is ow_last_error().  Everything a caller can get wrong is refused before any shard starts (OW_ERR_INVALID leaves no
 * trace).  There is no rollback beyond that: if a shard fails with OW_ERR_HIP / OW_ERR_NOMEM the other shards have already advanced
 * (their slices of `params` carry the new times) and the group is no longer in step -- destroy it and rebuild from the parameter
 * objects and the gathered normal maps (ow_set_normal_map), the same state a re-sharding uses. */
ow_status ow_group_update(ow_group *group, double delta, ow_cascade_params *params, int32_t count);
ow_status ow_group_process(ow_group *group);
ow_status ow_group_update_all(ow_group *group, double delta, ow_cascade_params *params, int32_t count);
ow_status ow_group_run(ow_group *group, double delta, ow_cascade_params *params, int32_t count, int32_t frames);
int32_t ow_group_cascades_remaining(const ow_group *group);
/* ow_sync of every shard (and of an outstanding gather) */
ow_status ow_group_sync(ow_group *group);

/* ow_group_gather_wait waits for EVERY shard's copy, then returns the first failure.  A shard whose kernels had reported a device-side
 * failure (the status word of ow_sync) when its layers landed makes the call return OW_ERR_HIP, and its layers of the gathered arrays
 * stay marked: ow_group_get_maps of those layers and ow_group_sample_surface over them keep returning OW_ERR_HIP until a later gather
 * of that shard has landed cleanly (the device pointers of ow_group_get_device_ptrs stay what they are: a caller that reads through
 * them takes gather_wait's status as the verdict on their contents). */
ow_status ow_group_gather_begin(ow_group *group);
ow_status ow_group_gather_wait(ow_group *group);
/* Duration (ms, begin of the first to end of the last copy, per shard, maximum over shards) and volume of the most recent completed
 * gather's inter-device copies; bytes_per_shard = cascades_per_device * N * N * 16. */
ow_status ow_group_gather_stats(ow_group *group, float *max_copy_ms, size_t *bytes_per_shard);

/* How a shard's layers reach the root device, as the HIP runtime reports it -- so that the first gather measured on a real node can be read
 * against the right model (bytes_per_shard / 153 GB/s over one xGMI link; a PCIe or staged path is several times slower):
 *   same_device  1: the shard sits on the root's device, its gather is a device-to-device copy in the shard's own stream;
 *   peer_access  hipDeviceCanAccessPeer(shard -> root): 1 = hipMemcpyPeerAsync is a direct write by the owning device's copy engine into the
 *                root's memory, 0 = the runtime stages it through the host;
 *   link_type    hipExtGetLinkTypeAndHopCount: 4 = xGMI, 2 = PCIe (HSA_AMD_LINK_INFO_TYPE_*: 0 HyperTransport, 1 QPI, 3 InfiniBand); -1 unknown;
 *   hops         ... its hop count (1 = a direct link); -1 unknown;
 *   staged_path  1: the shard gathers through snapshot + side stream + peer copy (every shard on another device, or all of them under
 *                OW_GROUP_FLAG_FORCE_PEER_PATH), 0: straight into the root's slots.
 * ow_query_link asks the same of any two device ordinals without a group (bench.py --gpus N prints it for every rank -> root pair). */
typedef struct ow_group_link {
    int32_t device, root_device;
    int32_t same_device, peer_access, link_type, hops, staged_path, reserved;
} ow_group_link;
#define OW_LINK_TYPE_PCIE 2
#define OW_LINK_TYPE_XGMI 4
ow_status ow_group_link_info(const ow_group *group, int32_t shard, ow_group_link *out);
ow_status ow_query_link(int32_t from_device, int32_t to_device, ow_group_link *out);

/* The gathered arrays on the root device (layout as ow_get_device_ptrs; layer g = global cascade g), as of the last gather. */
ow_status ow_group_get_device_ptrs(ow_group *group, void **displacement_map, void **normal_map, size_t *layer_stride_bytes);
/* Host copy of one gathered layer (as ow_get_maps); needs a completed gather (OW_ERR_STATE before the first one). */
ow_status ow_group_get_maps(ow_group *group, int32_t cascade, void *displacement_rgba16f, void *normal_rgba16f);
/* ow_sample_surface over the gathered arrays on the root device: what the consumer's shaders see (num_cascades <= 8 layers from 0). */
ow_status ow_group_sample_surface(ow_group *group, const float *world_xz, int32_t count, const float *map_scales, int32_t num_cascades,
                                  ow_surface_sample *out);

/* ---- zero-copy hand-off: the maps as dma-buf file descriptors ------------------------------------------------------ */

/* The reference's outputs are never copied: its compute shaders write the two array textures on the engine's own
 * RenderingDevice and the water shaders sample them in place (wave_generator.gd:19,34-35; README.md:85 -- the PCIe copy is
 * what killed the author's asynchronous experiment).  Across APIs the same needs shared memory, and this is the half of it
 * that is HIP's:
 *   ow_export_maps   : the context's displacement / normal arrays as two dma-buf file descriptors (the caller closes them).
 *                      A Vulkan consumer imports them with VkImportMemoryFdInfoKHR (VK_EXT_external_memory_dma_buf) into a
 *                      linear RGBA16F buffer / image of N x N x layers; another HIP process or context with ow_import_buffer.
 *                      A dma-buf covers a whole buffer object and the runtime packs allocations below 2 MiB into shared ones, so
 *                      the context allocates its arrays in multiples of 2 MiB (each descriptor maps its array from offset 0);
 *                      caller-owned arrays are exported only if they start a 2 MiB-aligned allocation of at least 2 MiB
 *                      (OW_ERR_STATE otherwise).
 *   ow_import_buffer : the other direction -- an fd exported elsewhere (VK_KHR_external_memory_fd from the engine's device,
 *                      or ow_export_maps in another process) becomes a device pointer to bytes [offset, offset + bytes) of that
 *                      memory on `device_id`, e.g. to be handed to
 *                      ow_create as ow_config.displacement_map / normal_map, so that the kernels write the engine's memory.
 *                      The fd stays the caller's (it is duplicated); ow_release_buffer unmaps.
 * Synchronisation stays with the caller (ow_sync / ow_readback-style fences before the consumer samples), as it does between
 * any two queues. */
typedef struct ow_imported ow_imported;
ow_status ow_export_maps(ow_context *ctx, int32_t *displacement_fd, int32_t *normal_fd, size_t *bytes_each);
ow_status ow_import_buffer(int32_t device_id, int32_t fd, size_t offset, size_t bytes, ow_imported **out, void **device_ptr);
void ow_release_buffer(ow_imported *imported);

/* ---- parity / debug ------------------------------------------------------------------------------ */

/* 8 FP32 channels per texel before FP16 quantisation: [hx, hy, hz, grad_x, grad_y, dhx_dx, foam, jacobian],
 * N*N*8 floats.  Requires OW_FLAG_DEBUG_F32. */
ow_status ow_get_maps_f32(ow_context *ctx, int32_t cascade, float *out);

/* The `spectrum` texture (wave_generator.gd:31; float4 = h0(k), conj(h0(-k)), N*N*4 floats) and the
 * FP32 dispersion plane omega(k) (N*N floats) the frame kernels consume.  Either may be NULL. */
ow_status ow_get_spectrum(ow_context *ctx, int32_t cascade, float *h0, float *omega);

/* The transposed intermediate after the first row pass, converted to the reference's layout
 * fft_buffer half 0 after transpose.glsl: [layer][row][col] complex, 4*N*N*2 floats.  The intermediate is scratch
 * shared by all batches: only cascades of the most recent pair of launches can be read (OW_ERR_STATE otherwise). */
ow_status ow_get_intermediate(ow_context *ctx, int32_t cascade, float *out);

/* The three push-constant blocks the reference packs for one _update of `cascade` (wave_generator.gd:71,73,85 through
 * RenderingContext.create_push_constant, render_context.gd:122-135: ints as s32, floats narrowed to f32, zero padding up to a
 * multiple of 16 bytes), as 32-bit words in the reference's own layouts, with the values this context's most recent launch for that
 * cascade was given.  This is where FP64 parameters become FP32: the parity tests hold these words bit for bit to the packing
 * restated from the reference.
 *   spectrum (spectrum_compute.glsl:18-30, 52 -> 64 bytes): seed.x, seed.y, tile_length.x, tile_length.y, alpha, peak_frequency,
 *            wind_speed, angle (rad), depth, swell, detail, spread, cascade_index -- of the most recent spectrum generation of this
 *            cascade (all zero before the first)
 *   modulate (spectrum_modulate.glsl:24-29, 20 -> 32 bytes): tile_length.x, tile_length.y, depth, time, cascade_index
 *   unpack   (fft_unpack.glsl:20-25, 16 bytes): cascade_index, whitecap, foam_grow_rate, foam_decay_rate
 * OW_ERR_STATE before the cascade's first launch. */
typedef struct ow_push_constants {
    uint32_t spectrum[16];
    uint32_t modulate[8];
    uint32_t unpack[4];
} ow_push_constants;
ow_status ow_get_push_constants(const ow_context *ctx, int32_t cascade, ow_push_constants *out);

/* ---- host math: static funcs of WaveGenerator (wave_generator.gd:116-121), FP64 ---------------------- */
double ow_jonswap_alpha(double wind_speed, double fetch_length_m);
double ow_jonswap_peak_angular_frequency(double wind_speed, double fetch_length_m);

/* ---- measurement ------------------------------------------------------------------------------------ */

/* Average duration (ms) of the two frame kernels over the launches made since the last reset, in situ: while
 * enabled, every launch carries start/stop hipEvents bound to its own dispatch packet (hipExtLaunchKernel), so the
 * figure is the kernel's begin -> end exactly as a rocprofv3 kernel trace reports it.  Throughput runs keep it off.
 * enable = 1: per pass -- ow_run stays on one launch per pass while enabled (ow_timing_read);
 * enable = 2: as launched -- ow_run keeps its tick groups / tick pairs and every such launch is timed (ow_timing_read_launches: average
 *             duration of those launches; the first and the last launch of a run carry one pass only), other launches as with 1. */
ow_status ow_timing_enable(ow_context *ctx, int32_t enable);
ow_status ow_timing_read(ow_context *ctx, float *pass1_ms_avg, float *pass2_ms_avg, int32_t *launches, int32_t reset);
ow_status ow_timing_read_launches(ow_context *ctx, float *launch_ms_avg, int32_t *launches, int32_t reset);

/* Kernel family the most recent batch was launched with: 1 = standard (k_pass1 / k_pass2), 2 = layer-parallel
 * (k_pass1_lp / k_pass2_lp), 3 = compact intermediate (k_pass1c / k_pass2c), 4 = layer-parallel on the compact intermediate (k_pass1c_lp /
 * k_pass2c_lp), 5 = that family launched in tick groups by ow_run (k_tick_group_c_lp), 6 = the compact family launched in tick pairs
 * by ow_run (k_tick_pair_c: pass 2 of one tick and pass 1 of the next in one launch); 0 before the first launch. */
int32_t ow_last_kernel_family(const ow_context *ctx);
/* How many consecutive ticks ow_run puts into one launch (tick groups, see OW_FLAG_NO_TICK_GROUPS): after an ow_run that went out in
 * tick groups or tick pairs (ow_last_kernel_family 5 / 6) the depth it used -- 1..8 for groups, limited by the scratch memory the
 * double-buffered intermediates take, 1 for pairs; otherwise the depth planned for this context's small batches (0: no tick groups). */
int32_t ow_tick_group_depth(const ow_context *ctx);
/* Number of cascades the most recent pair of launches processed (the runtime may split a tick into several pairs). */
int32_t ow_last_batch_cascades(const ow_context *ctx);

/* Benchmark probe: average duration (ms) of each frame kernel alone, from `reps` back-to-back launches of pass 1
 * and then `reps` of pass 2 with the arguments of the most recent batch, bracketed by hipEvents on the context's
 * stream (one pair of events per block of launches, so the event cost is amortised and the figure agrees with a
 * rocprofv3 kernel trace).  The extra pass-2 launches advance the foam recurrence: call it after a measurement,
 * never inside a simulation.  cascades_per_launch = how many cascades one launch of that batch covered. */
ow_status ow_probe_kernel_times(ow_context *ctx, int32_t reps, float *pass1_ms, float *pass2_ms, int32_t *cascades_per_launch);

/* Test hook.  bit 0, applied to the NEXT batch only: the second wave of every wave pair of the 2048^2 kernels never publishes its rendezvous
 * epoch, so its partner's bounded wait gives up -- exercises the status-word path above (a pending bit 0 keeps the look-ahead off).
 * bit 1, immediate: the device status word is set as a faulting launch would leave it, i.e. the failure is that of the launches IN FLIGHT
 * -- speculated pass-1 work included: the next synchronising call reports it, marks the layers enqueued since the last synchronisation and
 * drops whatever had been computed ahead.  Never set in normal operation. */
ow_status ow_debug_inject_fault(ow_context *ctx, uint32_t fault_bits);

/* Thread-local description of the last error returned on this thread ("" if none). */
const char *ow_last_error(void);
int32_t ow_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* OCEAN_WAVES_H */
