/* multi_gpu_host.c -- the C-ABI's device group from plain C: BASELINE config C4's shape (1024^2, one cascade per GPU, finished
 * layers gathered to the consumer's device) the way a single-process host (C#, GDExtension) would drive it.
 *   gcc -O2 -std=c99 -Iinclude examples/multi_gpu_host.c -o multi_gpu_host -Lgodotoceanwaves_amd -locean_waves \
 *       -Wl,-rpath,$PWD/godotoceanwaves_amd -Wl,-rpath-link,/opt/rocm/lib -lm
 *   ./multi_gpu_host [map_size [cascades_per_device [ticks [gather_every [dev0,dev1,... [peer]]]]]]      ("peer": every shard through the remote path, test hook)
 * Default device list: 0,0 (two shards on one device: exercises the whole path on a single-GPU box; OW_GROUP_FLAG_FORCE_PEER_PATH makes
 * both go through snapshot + side stream + hipMemcpyPeerAsync).  On a node: ./multi_gpu_host 1024 1 2000 16 0,1,2,3,4,5,6,7
 * Prints: maps/s without any gather, with a gather every `gather_every` ticks (overlapped), the copy time of one gather and the
 * bytes per shard, and a checksum of the gathered arrays after the run (the same on any device list: cascades are independent). */
#define _POSIX_C_SOURCE 199309L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "ocean_waves.h"

static double now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static uint64_t fnv1a(const void *data, size_t n, uint64_t h) {
    const unsigned char *p = (const unsigned char *)data;
    for (size_t i = 0; i < n; ++i) h = (h ^ p[i]) * 1099511628211ull;
    return h;
}
#define CHECK(call)                                                    \
    do {                                                               \
        if ((call) != OW_OK) {                                         \
            fprintf(stderr, "%s: %s\n", #call, ow_last_error());       \
            return 1;                                                  \
        }                                                              \
    } while (0)

/* rows 0-7 of SURVEY.md 8d's cascade table: tile, U, dir, fetch km, swell, spread, detail, whitecap, foam */
static const float kTable[8][9] = {{88, 10, 20, 150, 0.8f, 0.2f, 1, 0.5f, 8},  {57, 5, 15, 150, 0.8f, 0.4f, 1, 0.5f, 0},
                                   {16, 20, 20, 550, 0.8f, 0.4f, 1, 0.25f, 3}, {250, 20, 0, 550, 0.8f, 0.2f, 1, 0.5f, 5},
                                   {33, 20, 0, 550, 0.8f, 0.2f, 1, 0.5f, 5},   {137, 20, 0, 550, 0.8f, 0.2f, 1, 0.5f, 5},
                                   {23, 20, 0, 550, 0.8f, 0.2f, 1, 0.5f, 5},   {9, 20, 0, 550, 0.8f, 0.2f, 1, 0.5f, 5}};

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 256, per = argc > 2 ? atoi(argv[2]) : 1, ticks = argc > 3 ? atoi(argv[3]) : 200;
    const int gather_every = argc > 4 ? atoi(argv[4]) : 8;
    ow_group_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.map_size = n;
    cfg.cascades_per_device = per;
    cfg.depth = 20.0f;
    {   /* device list */
        char list[256];
        snprintf(list, sizeof list, "%s", argc > 5 ? argv[5] : "0,0");
        if (argc <= 5 || (argc > 6 && strcmp(argv[6], "peer") == 0)) cfg.flags |= OW_GROUP_FLAG_FORCE_PEER_PATH;
        for (char *tok = strtok(list, ","); tok && cfg.num_devices < OW_MAX_DEVICES; tok = strtok(NULL, ",")) cfg.device_ids[cfg.num_devices++] = atoi(tok);
    }
    cfg.root = 0;
    ow_group *g = NULL;
    CHECK(ow_group_create(&cfg, &g));
    const int total = ow_group_num_cascades(g);
    for (int s = 0; s < cfg.num_devices; ++s) {  /* how each shard's layers will reach the root: read the gather's figures against THIS */
        static const char *kLink[] = {"hypertransport", "qpi", "pcie", "infiniband", "xgmi"};
        ow_group_link l;
        CHECK(ow_group_link_info(g, s, &l));
        printf("shard %d: device %d -> root device %d: %s, peer_access=%d, link=%s, hops=%d, path=%s\n", s, l.device, l.root_device,
               l.same_device ? "same device" : "another device", l.peer_access, l.link_type >= 0 && l.link_type <= 4 ? kLink[l.link_type] : "unknown", l.hops,
               l.staged_path ? "snapshot + side stream + peer copy" : "device-to-device copy in the shard's stream");
    }
    ow_cascade_params *par = (ow_cascade_params *)calloc((size_t)total, sizeof *par);
    for (int i = 0; i < total; ++i) {
        const float *r = kTable[i % 8];
        ow_cascade_params_default(&par[i]);
        par[i].tile_length[0] = par[i].tile_length[1] = r[0];
        par[i].wind_speed = r[1]; par[i].wind_direction = r[2]; par[i].fetch_length = r[3];
        par[i].swell = r[4]; par[i].spread = r[5]; par[i].detail = r[6]; par[i].whitecap = r[7]; par[i].foam_amount = r[8];
        par[i].spectrum_seed[0] = 1000 + 17 * i; par[i].spectrum_seed[1] = -2000 + 31 * i;
        par[i].time = 120.0 + 3.14159265358979323846 * i;   /* water.gd:32 */
    }
    const double delta = 1.0 / 50.0;
    CHECK(ow_group_run(g, delta, par, total, 50));          /* spectra + warm-up */
    CHECK(ow_group_sync(g));

    double t0 = now();
    CHECK(ow_group_run(g, delta, par, total, ticks));
    CHECK(ow_group_sync(g));
    const double plain = now() - t0;

    t0 = now();
    for (int done = 0; done < ticks;) {                     /* a consumer that wants the arrays every `gather_every` ticks */
        const int k = ticks - done < gather_every ? ticks - done : gather_every;
        CHECK(ow_group_run(g, delta, par, total, k));
        CHECK(ow_group_gather_begin(g));                    /* snapshot in stream order; the copies overlap the next ticks */
        done += k;
    }
    CHECK(ow_group_gather_wait(g));
    CHECK(ow_group_sync(g));
    const double gathered = now() - t0;
    float copy_ms = 0.0f;
    size_t shard_bytes = 0;
    CHECK(ow_group_gather_stats(g, &copy_ms, &shard_bytes));

    const size_t bytes = (size_t)n * n * 8;
    void *d = malloc(bytes), *m = malloc(bytes);
    uint64_t sum = 1469598103934665603ull;
    for (int c = 0; c < total; ++c) {
        CHECK(ow_group_get_maps(g, c, d, m));
        sum = fnv1a(m, bytes, fnv1a(d, bytes, sum));
    }
    {   /* the model the measurement is to be read against: cascades share nothing, so without the exchange the group's rate is the
         * sum of its devices' rates; a shard's two maps (16 B/texel) cross ONE xGMI link to the consumer (~153 GB/s per link, point to point:
         * the root's seven inbound links carry one sender each), and the gather hides under the ticks in between while
         * gather_every >= 1.25 x (link time / tick time). */
        const double tick_ms = plain / ticks * 1e3, link_ms = 16.0 * n * n * per / 153e9 * 1e3;
        int k_model = (int)(1.25 * link_ms / tick_ms) + 1;
        printf("model: per-device tick %.4f ms (measured, no gather); one shard's gather = %zu bytes = %.4f ms at 153 GB/s per link; the gather hides "
               "under compute from gather_every >= %d; expected maps_per_s with gather = maps_per_s_no_gather while gather_every (%d) >= that, else "
               "bounded by the links at %.1f maps/s\n",
               tick_ms, (size_t)16 * n * n * per, link_ms, k_model, gather_every, (double)total * gather_every / (link_ms * 1e-3));
    }
    printf("devices=%d cascades=%d map_size=%d ticks=%d maps_per_s_no_gather=%.1f maps_per_s_gather_every_%d=%.1f gather_copy_ms=%.4f "
           "bytes_per_shard=%zu link_GBps_per_shard=%.2f checksum=%016llx time=%.17g\n",
           cfg.num_devices, total, n, ticks, (double)total * ticks / plain, gather_every, (double)total * ticks / gathered, (double)copy_ms,
           shard_bytes, copy_ms > 0 ? (double)shard_bytes / (copy_ms * 1e-3) / 1e9 : 0.0, (unsigned long long)sum, par[total - 1].time);
    free(d);
    free(m);
    free(par);
    ow_group_destroy(g);
    return 0;
}
