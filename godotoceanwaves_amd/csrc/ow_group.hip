// ow_group.hip -- cascades sharded over the GPUs of one node, inside one process (include/ocean_waves.h "several devices").
//
// The reference has one device and says only that cascades are independent (assets/water/wave_generator.gd:65-85 touches nothing of
// another cascade; README.md:77-80).  Here that independence is the partition: shard s = one ow_context on device_ids[s] owning a
// contiguous block of global cascades.  No data-path exchange exists; the one exchange is the gather of finished RGBA16F layers into
// the consumer's two array textures on the root device (what water.gd:95-100 binds), and only owned layers travel:
//
//   shard stream :  ... ticks ... | snapshot (D2D, 16 B/texel per owned layer) | ticks continue ...
//   copy stream  :                     wait(snapshot) | hipMemcpyPeerAsync -> root's layer slots (xGMI) | event
//
// Each shard has a worker thread: a C / C# host calls ow_group_run once and the N devices are fed side by side (a single thread
// walking eight devices would be enqueue-bound at 1024^2 x 1: ~15 us of kernel per tick against 8 x ~5 us of launch calls).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "ow_internal.h"
#include "ow_kernels.h"

namespace {

using ow::fail;

// one worker thread per shard: runs the tasks it is handed, in order; the caller waits for the result
struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<ow_status()> task;
    bool has_task = false, done = false, quit = false;
    ow_status status = OW_OK;
    std::string message;

    void start() {
        th = std::thread([this] {
            std::unique_lock<std::mutex> lk(mu);
            for (;;) {
                cv.wait(lk, [this] { return has_task || quit; });
                if (quit) return;
                auto fn = std::move(task);
                has_task = false;
                lk.unlock();
                const ow_status st = fn();
                std::string msg = st != OW_OK ? ow_last_error() : "";
                lk.lock();
                status = st;
                message = std::move(msg);
                done = true;
                cv.notify_all();
            }
        });
    }
    void post(std::function<ow_status()> fn) {
        std::lock_guard<std::mutex> lk(mu);
        task = std::move(fn);
        has_task = true;
        done = false;
        cv.notify_all();
    }
    ow_status wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return done; });
        return status;
    }
    void stop() {
        if (!th.joinable()) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
            cv.notify_all();
        }
        th.join();
    }
};

struct Shard {
    int device = 0;
    bool remote = false;  // gathers through snapshot + side stream + peer copy (false: on the root's device, copies straight into its slots)
    ow_context *ctx = nullptr;
    hipStream_t stream = nullptr, copy_stream = nullptr;
    char *disp = nullptr, *norm = nullptr;  // the shard's live maps
    char *snap = nullptr;                   // [2 maps][owned layers][N][N] RGBA16F (remote shards)
    hipEvent_t snap_ready = nullptr, copy_start = nullptr, copy_done = nullptr;
    bool pending = false;  // a gather has been begun and not yet waited for
    ow_group_link link{};  // how this shard's layers reach the root (ow_group_link_info), probed once by ow_group_create
    Worker worker;
};

}  // namespace

struct ow_group {
    int n = 0, shards = 0, per = 0, root = 0, total = 0, layers = 0, root_device = 0;
    size_t plane_bytes = 0;  // N * N * 8: one layer of one map
    Shard s[OW_MAX_DEVICES];
    char *gdisp = nullptr, *gnorm = nullptr;  // gathered arrays on the root device: [layers][N][N] RGBA16F
    bool own_gdisp = false, own_gnorm = false;
    hipStream_t root_stream = nullptr;  // ow_group_get_maps / ow_group_sample_surface
    bool gathered = false;              // at least one gather has completed
    // shards (bit i) whose layers of the gathered arrays are NOT maps: the shard's kernels had reported a device-side failure when its
    // copy landed.  Set by gather_wait_all, lifted by the next gather of that shard that lands cleanly; while a bit is set,
    // ow_group_get_maps of that shard's layers and ow_group_sample_surface over them are refused (the context-level contract of
    // ow_sync, carried over to the group's readers).
    uint32_t faulted_shards = 0;
    float last_copy_ms = 0.0f;
    float *query_xz = nullptr;
    ow::SurfaceSample *query_out = nullptr;
    int query_capacity = 0;
};

namespace {

// run fn(shard index) on every shard's worker; the first failure (lowest shard) is the group's status and message
ow_status on_all(ow_group *g, const std::function<ow_status(int)> &fn) {
    for (int i = 0; i < g->shards; ++i) g->s[i].worker.post([&fn, i] { return fn(i); });
    ow_status first = OW_OK;
    for (int i = 0; i < g->shards; ++i) {
        const ow_status st = g->s[i].worker.wait();
        if (st != OW_OK && first == OW_OK) {
            first = st;
            ow::set_last_error(("shard " + std::to_string(i) + " (device " + std::to_string(g->s[i].device) + "): " + g->s[i].worker.message).c_str());
        }
    }
    return first;
}
ow_status on_one(ow_group *g, int i, const std::function<ow_status()> &fn) {
    g->s[i].worker.post(fn);
    const ow_status st = g->s[i].worker.wait();
    if (st != OW_OK) ow::set_last_error(("shard " + std::to_string(i) + " (device " + std::to_string(g->s[i].device) + "): " + g->s[i].worker.message).c_str());
    return st;
}

// All records are checked before any shard starts, so that a refused call leaves no shard a tick ahead of the others.
ow_status check_records(const ow_group *g, const ow_cascade_params *params, int count, double delta) {
    if (!g || !params) return fail(OW_ERR_INVALID, "null argument");
    if (count != g->total) return fail(OW_ERR_INVALID, "count %d: a group call takes the records of all %d cascades", count, g->total);
    return ow::validate_records(params, count, delta);
}

bool gather_in_flight(const ow_group *g) {
    for (int i = 0; i < g->shards; ++i)
        if (g->s[i].pending) return true;
    return false;
}

// one shard's part of a gather, on its worker thread
ow_status gather_begin_shard(ow_group *g, int i) {
    Shard &sh = g->s[i];
    OW_HIP(hipSetDevice(sh.device));
    const size_t bytes = (size_t)g->per * g->plane_bytes, off = (size_t)i * g->per * g->plane_bytes;
    if (!sh.remote) {  // same device as the consumer: the copy into the slots IS the snapshot, in the shard's stream order
        OW_HIP(hipEventRecord(sh.copy_start, sh.stream));
        OW_HIP(hipMemcpyAsync(g->gdisp + off, sh.disp, bytes, hipMemcpyDeviceToDevice, sh.stream));
        OW_HIP(hipMemcpyAsync(g->gnorm + off, sh.norm, bytes, hipMemcpyDeviceToDevice, sh.stream));
        OW_HIP(hipEventRecord(sh.copy_done, sh.stream));
        sh.pending = true;
        return OW_OK;
    }
    // the snapshot buffer may still be feeding the previous gather's peer copy
    if (sh.pending) OW_HIP(hipStreamWaitEvent(sh.stream, sh.copy_done, 0));
    OW_HIP(hipMemcpyAsync(sh.snap, sh.disp, bytes, hipMemcpyDeviceToDevice, sh.stream));
    OW_HIP(hipMemcpyAsync(sh.snap + bytes, sh.norm, bytes, hipMemcpyDeviceToDevice, sh.stream));
    OW_HIP(hipEventRecord(sh.snap_ready, sh.stream));
    OW_HIP(hipStreamWaitEvent(sh.copy_stream, sh.snap_ready, 0));
    OW_HIP(hipEventRecord(sh.copy_start, sh.copy_stream));
    // pushed by the owning device (its copy engine writes across the link); only the owned layers travel
    OW_HIP(hipMemcpyPeerAsync(g->gdisp + off, g->root_device, sh.snap, sh.device, bytes, sh.copy_stream));
    OW_HIP(hipMemcpyPeerAsync(g->gnorm + off, g->root_device, sh.snap + bytes, sh.device, bytes, sh.copy_stream));
    OW_HIP(hipEventRecord(sh.copy_done, sh.copy_stream));
    sh.pending = true;
    return OW_OK;
}

// Every pending shard is waited for and polled, whatever happens to the others (a failure of shard 2 must not leave shards 3.. pending
// for ever); the FIRST failure is what the call returns.
ow_status gather_wait_all(ow_group *g) {
    float worst = 0.0f;
    bool any = false;
    ow_status first = OW_OK;
    std::string first_message;
    auto note = [&](int i, ow_status st) {
        g->faulted_shards |= 1u << i;
        if (first != OW_OK) return;
        first = st;
        first_message = "shard " + std::to_string(i) + " (device " + std::to_string(g->s[i].device) + "): " + ow_last_error();
    };
    for (int i = 0; i < g->shards; ++i) {
        Shard &sh = g->s[i];
        if (!sh.pending) continue;
        sh.pending = false;
        hipError_t e = hipSetDevice(sh.device);
        if (e == hipSuccess) e = hipEventSynchronize(sh.copy_done);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            note(i, fail(OW_ERR_HIP, "waiting for the gather's copy failed: %s", hipGetErrorString(e)));
            continue;
        }
        any = true;
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, sh.copy_start, sh.copy_done) == hipSuccess) worst = std::max(worst, ms);
        else (void)hipGetLastError();
        // what has landed is only maps if the shard's kernels reported no failure (the copy was ordered behind them)
        if (ow_status st = ow::poll_status(sh.ctx); st != OW_OK) note(i, st);
        else g->faulted_shards &= ~(1u << i);  // this shard's layers are maps again
    }
    if (any) {
        g->last_copy_ms = worst;
        g->gathered = true;
    }
    if (first != OW_OK) ow::set_last_error(first_message.c_str());
    return first;
}
// the readers' side of it: layers [first, first + count) of the gathered arrays
ow_status refuse_faulted_layers(const ow_group *g, int first, int count) {
    for (int layer = first; layer < first + count && layer < g->total; ++layer)
        if ((g->faulted_shards >> (layer / g->per)) & 1u)
            return fail(OW_ERR_HIP, "gathered layer %d belongs to shard %d, whose kernels had reported a device-side failure when its layers were "
                                    "gathered: the bytes are not maps until a later gather of that shard has landed cleanly", layer, layer / g->per);
    return OW_OK;
}

}  // namespace

extern "C" {

// what lies between two devices, as the runtime reports it: peer access (the owner's copy engine writes the consumer's memory directly) and the
// link (hipExtGetLinkTypeAndHopCount: 4 = xGMI, 2 = PCIe) -- so that a measured gather can be read against the right model
ow_status ow_query_link(int32_t from_device, int32_t to_device, ow_group_link *out) {
    if (!out) return fail(OW_ERR_INVALID, "null argument");
    std::memset(out, 0, sizeof(*out));
    out->device = from_device;
    out->root_device = to_device;
    out->link_type = out->hops = -1;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(OW_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (from_device < 0 || from_device >= ndev || to_device < 0 || to_device >= ndev) return fail(OW_ERR_INVALID, "device %d or %d outside [0,%d)", from_device, to_device, ndev);
    if (from_device == to_device) {
        out->same_device = 1;
        out->peer_access = 1;
        out->hops = 0;
        return OW_OK;
    }
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, from_device, to_device) == hipSuccess) out->peer_access = can ? 1 : 0;
    else (void)hipGetLastError();
    uint32_t type = 0, hops = 0;
    if (hipExtGetLinkTypeAndHopCount(from_device, to_device, &type, &hops) == hipSuccess) {
        out->link_type = (int32_t)type;
        out->hops = (int32_t)hops;
    } else {
        (void)hipGetLastError();
    }
    return OW_OK;
}

ow_status ow_group_link_info(const ow_group *g, int32_t shard, ow_group_link *out) {
    if (!g || !out) return fail(OW_ERR_INVALID, "null argument");
    if (shard < 0 || shard >= g->shards) return fail(OW_ERR_INVALID, "shard %d outside [0,%d)", shard, g->shards);
    *out = g->s[shard].link;
    return OW_OK;
}

ow_status ow_group_create(const ow_group_config *cfg, ow_group **out) {
    if (!cfg || !out) return fail(OW_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->num_devices < 1 || cfg->num_devices > OW_MAX_DEVICES) return fail(OW_ERR_INVALID, "num_devices %d outside [1,%d]", cfg->num_devices, OW_MAX_DEVICES);
    if (cfg->cascades_per_device < 1 || cfg->cascades_per_device > OW_MAX_CASCADES)
        return fail(OW_ERR_INVALID, "cascades_per_device %d outside [1,%d]", cfg->cascades_per_device, OW_MAX_CASCADES);
    if (cfg->root < 0 || cfg->root >= cfg->num_devices) return fail(OW_ERR_INVALID, "root %d is not an index into the %d device_ids", cfg->root, cfg->num_devices);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(OW_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    for (int i = 0; i < cfg->num_devices; ++i)
        if (cfg->device_ids[i] < 0 || cfg->device_ids[i] >= ndev) return fail(OW_ERR_INVALID, "device_ids[%d] = %d outside [0,%d)", i, cfg->device_ids[i], ndev);
    int caller_dev = 0;
    OW_HIP(hipGetDevice(&caller_dev));
    struct DeviceRestore {
        int dev;
        ~DeviceRestore() { (void)hipSetDevice(dev); }
    } restore{caller_dev};

    ow_group *g = new (std::nothrow) ow_group();
    if (!g) return fail(OW_ERR_NOMEM, "out of host memory");
    g->n = cfg->map_size;
    g->shards = cfg->num_devices;
    g->per = cfg->cascades_per_device;
    g->root = cfg->root;
    g->total = g->shards * g->per;
    g->layers = std::max(2, g->total);  // init_gpu(maxi(2, n)), water.gd:91
    g->root_device = cfg->device_ids[cfg->root];
    g->plane_bytes = (size_t)g->n * g->n * 8;
    auto bail = [&](ow_status st) {
        const std::string keep = ow_last_error();
        ow_group_destroy(g);
        ow::set_last_error(keep.c_str());
        return st;
    };
    const bool force_peer = (cfg->flags & OW_GROUP_FLAG_FORCE_PEER_PATH) != 0;
    for (int i = 0; i < g->shards; ++i) {
        Shard &sh = g->s[i];
        sh.device = cfg->device_ids[i];
        sh.remote = force_peer || sh.device != g->root_device;
        (void)ow_query_link(sh.device, g->root_device, &sh.link);  // (a probe: what it cannot find out stays -1)
        sh.link.staged_path = sh.remote ? 1 : 0;
        if (hipSetDevice(sh.device) != hipSuccess) return bail(fail(OW_ERR_HIP, "hipSetDevice(%d) failed", sh.device));
        if (sh.device != g->root_device) {  // let the owning device write into the root's memory directly (xGMI); without peer access
            int can = 0;                    // hipMemcpyPeerAsync still works, staged by the runtime
            if (hipDeviceCanAccessPeer(&can, sh.device, g->root_device) == hipSuccess && can) {
                const hipError_t e = hipDeviceEnablePeerAccess(g->root_device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return bail(fail(OW_ERR_HIP, "hipDeviceEnablePeerAccess(%d -> %d): %s", sh.device, g->root_device, hipGetErrorString(e)));
                (void)hipGetLastError();
            }
        }
        if (hipStreamCreateWithFlags(&sh.stream, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&sh.copy_stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&sh.snap_ready, hipEventDisableTiming) != hipSuccess || hipEventCreate(&sh.copy_start) != hipSuccess ||
            hipEventCreate(&sh.copy_done) != hipSuccess)
            return bail(fail(OW_ERR_HIP, "stream / event creation failed on device %d", sh.device));
        if (sh.remote && hipMalloc((void **)&sh.snap, 2 * (size_t)g->per * g->plane_bytes) != hipSuccess)
            return bail(fail(OW_ERR_NOMEM, "hipMalloc of the %zu-byte gather snapshot failed on device %d", 2 * (size_t)g->per * g->plane_bytes, sh.device));
        ow_config c;
        std::memset(&c, 0, sizeof(c));
        c.map_size = cfg->map_size;
        c.num_cascades = g->per;
        c.device_id = sh.device;
        c.depth = cfg->depth;
        c.stream = sh.stream;
        c.flags = cfg->flags & 0xFFFFu;
        if (ow_status st = ow_create(&c, &sh.ctx); st != OW_OK) return bail(st);
        void *d = nullptr, *nm = nullptr;
        if (ow_status st = ow_get_device_ptrs(sh.ctx, &d, &nm, nullptr); st != OW_OK) return bail(st);
        sh.disp = (char *)d;
        sh.norm = (char *)nm;
        sh.worker.start();
    }
    if (hipSetDevice(g->root_device) != hipSuccess) return bail(fail(OW_ERR_HIP, "hipSetDevice(%d) failed", g->root_device));
    const size_t array_bytes = (size_t)g->layers * g->plane_bytes;
    if (cfg->displacement_map) g->gdisp = (char *)cfg->displacement_map;
    else if (hipMalloc((void **)&g->gdisp, array_bytes) == hipSuccess) g->own_gdisp = true;
    else return bail(fail(OW_ERR_NOMEM, "hipMalloc of the %zu-byte gathered displacement array failed", array_bytes));
    if (cfg->normal_map) g->gnorm = (char *)cfg->normal_map;
    else if (hipMalloc((void **)&g->gnorm, array_bytes) == hipSuccess) g->own_gnorm = true;
    else return bail(fail(OW_ERR_NOMEM, "hipMalloc of the %zu-byte gathered normal array failed", array_bytes));
    if (hipStreamCreateWithFlags(&g->root_stream, hipStreamNonBlocking) != hipSuccess || hipMemsetAsync(g->gdisp, 0, array_bytes, g->root_stream) != hipSuccess ||
        hipMemsetAsync(g->gnorm, 0, array_bytes, g->root_stream) != hipSuccess || hipStreamSynchronize(g->root_stream) != hipSuccess)
        return bail(fail(OW_ERR_HIP, "initialising the gathered arrays failed"));
    *out = g;
    return OW_OK;
}

void ow_group_destroy(ow_group *g) {
    if (!g) return;
    int caller_dev = -1;
    (void)hipGetDevice(&caller_dev);
    for (int i = 0; i < g->shards; ++i) g->s[i].worker.stop();
    for (int i = 0; i < g->shards; ++i) {
        Shard &sh = g->s[i];
        (void)hipSetDevice(sh.device);
        if (sh.copy_stream) (void)hipStreamSynchronize(sh.copy_stream);
        if (sh.stream) (void)hipStreamSynchronize(sh.stream);
        ow_destroy(sh.ctx);  // (borrowed stream: the context does not destroy it)
        (void)hipFree(sh.snap);
        for (hipEvent_t e : {sh.snap_ready, sh.copy_start, sh.copy_done})
            if (e) (void)hipEventDestroy(e);
        if (sh.copy_stream) (void)hipStreamDestroy(sh.copy_stream);
        if (sh.stream) (void)hipStreamDestroy(sh.stream);
    }
    (void)hipSetDevice(g->root_device);
    if (g->root_stream) {
        (void)hipStreamSynchronize(g->root_stream);
        (void)hipStreamDestroy(g->root_stream);
    }
    if (g->own_gdisp) (void)hipFree(g->gdisp);
    if (g->own_gnorm) (void)hipFree(g->gnorm);
    (void)hipFree(g->query_xz);
    (void)hipFree(g->query_out);
    delete g;
    if (caller_dev >= 0) (void)hipSetDevice(caller_dev);
}

int32_t ow_group_num_cascades(const ow_group *g) { return g ? g->total : 0; }

ow_context *ow_group_context(ow_group *g, int32_t shard) { return (g && shard >= 0 && shard < g->shards) ? g->s[shard].ctx : nullptr; }

ow_status ow_group_update(ow_group *g, double delta, ow_cascade_params *params, int32_t count) {
    if (ow_status st = check_records(g, params, count, delta); st != OW_OK) return st;
    return on_all(g, [&](int i) { return ow_update(g->s[i].ctx, delta, params + (size_t)i * g->per, g->per); });
}

ow_status ow_group_update_all(ow_group *g, double delta, ow_cascade_params *params, int32_t count) {
    if (ow_status st = check_records(g, params, count, delta); st != OW_OK) return st;
    return on_all(g, [&](int i) { return ow_update_all(g->s[i].ctx, delta, params + (size_t)i * g->per, g->per); });
}

ow_status ow_group_run(ow_group *g, double delta, ow_cascade_params *params, int32_t count, int32_t frames) {
    if (ow_status st = check_records(g, params, count, delta); st != OW_OK) return st;
    return on_all(g, [&](int i) { return ow_run(g->s[i].ctx, delta, params + (size_t)i * g->per, g->per, frames); });
}

int32_t ow_group_cascades_remaining(const ow_group *g) {
    int total = 0;
    if (g)
        for (int i = 0; i < g->shards; ++i) total += ow_cascades_remaining(g->s[i].ctx);
    return total;
}

ow_status ow_group_process(ow_group *g) {  // wave_generator.gd:56-63: one armed cascade, highest (global) index first
    if (!g) return fail(OW_ERR_INVALID, "null group");
    for (int i = g->shards - 1; i >= 0; --i)
        if (ow_cascades_remaining(g->s[i].ctx) > 0) return on_one(g, i, [g, i] { return ow_process(g->s[i].ctx); });
    return OW_OK;  // nothing armed: a no-op, like the reference
}

ow_status ow_group_sync(ow_group *g) {
    if (!g) return fail(OW_ERR_INVALID, "null group");
    const ow_status st = on_all(g, [&](int i) { return ow_sync(g->s[i].ctx); });
    const std::string keep = st != OW_OK ? ow_last_error() : "";
    for (int i = 0; i < g->shards; ++i) {  // an outstanding gather is part of "everything enqueued"
        Shard &sh = g->s[i];
        if (!sh.pending) continue;
        (void)hipSetDevice(sh.device);
        if (hipEventSynchronize(sh.copy_done) != hipSuccess && st == OW_OK) return fail(OW_ERR_HIP, "waiting for shard %d's gather failed", i);
    }
    if (st != OW_OK) ow::set_last_error(keep.c_str());
    return st;
}

ow_status ow_group_gather_begin(ow_group *g) {
    if (!g) return fail(OW_ERR_INVALID, "null group");
    return on_all(g, [&](int i) { return gather_begin_shard(g, i); });
}

ow_status ow_group_gather_wait(ow_group *g) {
    if (!g) return fail(OW_ERR_INVALID, "null group");
    bool any = false;
    for (int i = 0; i < g->shards; ++i) any = any || g->s[i].pending;
    if (!any) return fail(OW_ERR_STATE, "no gather is outstanding");
    int caller_dev = -1;
    (void)hipGetDevice(&caller_dev);
    const ow_status st = gather_wait_all(g);
    if (caller_dev >= 0) (void)hipSetDevice(caller_dev);
    return st;
}

ow_status ow_group_gather_stats(ow_group *g, float *max_copy_ms, size_t *bytes_per_shard) {
    if (!g) return fail(OW_ERR_INVALID, "null group");
    if (!g->gathered) return fail(OW_ERR_STATE, "no gather has completed yet");
    if (max_copy_ms) *max_copy_ms = g->last_copy_ms;
    if (bytes_per_shard) *bytes_per_shard = 2 * (size_t)g->per * g->plane_bytes;
    return OW_OK;
}

ow_status ow_group_get_device_ptrs(ow_group *g, void **disp, void **norm, size_t *stride) {
    if (!g) return fail(OW_ERR_INVALID, "null group");
    if (disp) *disp = g->gdisp;
    if (norm) *norm = g->gnorm;
    if (stride) *stride = g->plane_bytes;
    return OW_OK;
}

ow_status ow_group_get_maps(ow_group *g, int32_t cascade, void *disp, void *norm) {
    if (!g) return fail(OW_ERR_INVALID, "null group");
    if (cascade < 0 || cascade >= g->layers) return fail(OW_ERR_INVALID, "cascade %d out of range [0,%d)", cascade, g->layers);
    if (!g->gathered) return fail(OW_ERR_STATE, "the gathered arrays are empty: ow_group_gather_begin / ow_group_gather_wait first");
    if (gather_in_flight(g)) return fail(OW_ERR_STATE, "a gather is in flight (the arrays are being written): ow_group_gather_wait first");
    if (ow_status st = refuse_faulted_layers(g, cascade, 1); st != OW_OK) return st;
    int caller_dev = -1;
    (void)hipGetDevice(&caller_dev);
    auto run = [&]() -> ow_status {
        OW_HIP(hipSetDevice(g->root_device));
        if (disp) OW_HIP(hipMemcpyAsync(disp, g->gdisp + (size_t)cascade * g->plane_bytes, g->plane_bytes, hipMemcpyDeviceToHost, g->root_stream));
        if (norm) OW_HIP(hipMemcpyAsync(norm, g->gnorm + (size_t)cascade * g->plane_bytes, g->plane_bytes, hipMemcpyDeviceToHost, g->root_stream));
        OW_HIP(hipStreamSynchronize(g->root_stream));
        return OW_OK;
    };
    const ow_status st = run();
    if (caller_dev >= 0) (void)hipSetDevice(caller_dev);
    return st;
}

ow_status ow_group_sample_surface(ow_group *g, const float *xz, int32_t count, const float *map_scales, int32_t num_cascades, ow_surface_sample *out) {
    static_assert(sizeof(ow_surface_sample) == sizeof(ow::SurfaceSample), "record layout");
    if (!g) return fail(OW_ERR_INVALID, "null group");
    if (count < 0) return fail(OW_ERR_INVALID, "count must be >= 0");
    if (num_cascades < 1 || num_cascades > std::min(g->total, OW_MAX_CASCADES))
        return fail(OW_ERR_INVALID, "num_cascades %d outside [1,%d]", num_cascades, std::min(g->total, OW_MAX_CASCADES));
    if (!g->gathered) return fail(OW_ERR_STATE, "the gathered arrays are empty: ow_group_gather_begin / ow_group_gather_wait first");
    if (gather_in_flight(g)) return fail(OW_ERR_STATE, "a gather is in flight (the arrays are being written): ow_group_gather_wait first");
    if (ow_status st = refuse_faulted_layers(g, 0, num_cascades); st != OW_OK) return st;
    if (count == 0) return OW_OK;
    if (!xz || !map_scales || !out) return fail(OW_ERR_INVALID, "null argument");
    int caller_dev = -1;
    (void)hipGetDevice(&caller_dev);
    auto run = [&]() -> ow_status {
        OW_HIP(hipSetDevice(g->root_device));
        if (count > g->query_capacity) {
            (void)hipFree(g->query_xz);
            (void)hipFree(g->query_out);
            g->query_xz = nullptr;
            g->query_out = nullptr;
            g->query_capacity = 0;
            const int cap = std::max(count, 4096);
            if (hipMalloc((void **)&g->query_xz, (size_t)cap * 2 * sizeof(float)) != hipSuccess || hipMalloc((void **)&g->query_out, (size_t)cap * sizeof(ow::SurfaceSample)) != hipSuccess)
                return fail(OW_ERR_NOMEM, "hipMalloc failed for %d query points", cap);
            g->query_capacity = cap;
        }
        ow::SurfaceScales sc;
        std::memset(&sc, 0, sizeof(sc));
        std::memcpy(sc.s, map_scales, (size_t)num_cascades * 4 * sizeof(float));
        ow::DeviceBuffers buf;
        std::memset(&buf, 0, sizeof(buf));
        buf.disp = (ow::u16x4 *)g->gdisp;  // the sampling kernel reads the two array textures only
        buf.norm = (ow::u16x4 *)g->gnorm;
        OW_HIP(hipMemcpyAsync(g->query_xz, xz, (size_t)count * 2 * sizeof(float), hipMemcpyHostToDevice, g->root_stream));
        OW_HIP(ow::launch_sample_surface(g->n, num_cascades, buf, g->query_xz, count, sc, g->query_out, g->root_stream));
        OW_HIP(hipMemcpyAsync(out, g->query_out, (size_t)count * sizeof(ow::SurfaceSample), hipMemcpyDeviceToHost, g->root_stream));
        OW_HIP(hipStreamSynchronize(g->root_stream));
        return OW_OK;
    };
    const ow_status st = run();
    if (caller_dev >= 0) (void)hipSetDevice(caller_dev);
    return st;
}

}  // extern "C"
