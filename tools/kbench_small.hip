// kbench_small.hip -- developer microbenchmark (not part of the product): where the time of a SMALL tick goes
// (256^2 / 512^2 x 1..4 cascades: the layer-parallel kernels on the compact intermediate, k_pass1c_lp / k_pass2c_lp).
//   * tick and per-kernel durations (hipEvents around back-to-back launches);
//   * the launch floor: an empty kernel and a "one load, one store" kernel with the same grids;
//   * per-wave phase stamps of both kernels (STAMPS instantiation), averaged over the waves that did work, plus the
//     spread of wave start / end times over the launch (dispatch ramp, tail).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DKS_N=256 -I godotoceanwaves_amd/csrc tools/kbench_small.hip -o tools/kbench_small_256
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ow_frame_kernels.h"
#include "ow_tables.h"

#ifndef KS_N
#define KS_N 256
#endif
using namespace ow;
#define CK(x)                                                                                \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) {                                                              \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);   \
            exit(1);                                                                         \
        }                                                                                    \
    } while (0)

__global__ void k_empty() {}
__global__ void k_touch(const float *in, float *out) {  // one dependent load -> store per wave: a memory round trip and nothing else
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = in[(blockIdx.x * 97 + threadIdx.x) & 0xFFFF] + 1.0f;
}

template <class F>
static float time_it(F f, int iters, hipStream_t s) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 20; ++i) f();
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters * 1e3f;
}

static void report(const char *title, const std::vector<Stamp> &h, const char *const *names, int nnames, float kernel_us) {
    unsigned long long t0 = ~0ull, t1 = 0, last_start = 0, first_end = ~0ull;
    int cnt = 0;
    for (auto &x : h) {
        if (!x.t[0]) continue;
        ++cnt;
        t0 = std::min(t0, x.t[0]);
        last_start = std::max(last_start, x.t[0]);
        t1 = std::max(t1, x.t[14]);
        first_end = std::min(first_end, x.t[14]);
    }
    if (!cnt) {
        printf("%s: no stamps\n", title);
        return;
    }
    const double span = (double)(t1 - t0), per_us = span / kernel_us;
    printf("%s: %d working waves; first wave start -> last wave end = %.0f ticks (kernel %.2f us by events => %.0f ticks/us)\n", title, cnt, span, kernel_us, per_us);
    printf("  wave starts spread over %.0f ticks (%.2f us), wave ends spread over %.0f ticks (%.2f us)\n", (double)(last_start - t0),
           (last_start - t0) / per_us, (double)(t1 - first_end), (t1 - first_end) / per_us);
    double prev = 0;
    for (int k = 1; k < 15; ++k) {
        if (k >= nnames || !names[k][0]) continue;
        double avg = 0;
        int c2 = 0;
        for (auto &x : h)
            if (x.t[0] && x.t[k]) {
                avg += (double)(x.t[k] - x.t[0]);
                ++c2;
            }
        if (!c2) continue;
        avg /= c2;
        printf("  %-44s %9.0f ticks  (+%6.0f = %5.2f us)   [%d waves]\n", names[k], avg, avg - prev, (avg - prev) / per_us, c2);
        prev = avg;
    }
}

int main(int argc, char **argv) {
    constexpr int N = KS_N;
    const int C = argc > 1 ? atoi(argv[1]) : 4, iters = argc > 2 ? atoi(argv[2]) : 2000;
    const size_t pl = (size_t)N * N, L = C;
    DeviceBuffers buf{};
    CK(hipMalloc((void **)&buf.h0, L * pl * 8));
    CK(hipMalloc(&buf.omega, L * pl * 4));
    CK(hipMalloc((void **)&buf.T, 2 * L * pl * 32));  // twice: the tick loop double-buffers it
    CK(hipMalloc(&buf.disp, L * pl * 8));
    CK(hipMalloc(&buf.norm, L * pl * 8));
    CK(hipMalloc(&buf.foam, L * pl * 2));
    CK(hipMemset(buf.foam, 0, L * pl * 2));
    CK(hipMalloc((void **)&buf.pcol, (size_t)2 * C * N * 8));
    CK(hipMalloc((void **)&buf.rrow, (size_t)2 * C * N * 32));
    CK(hipMalloc((void **)&buf.status, 64));
    CK(hipMemset(buf.status, 0, 64));
    std::vector<float> hh(L * pl * 2);
    for (size_t i = 0; i < hh.size(); ++i) hh[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f - 0.5f;
    CK(hipMemcpy(buf.h0, hh.data(), hh.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> om(L * pl);
    for (size_t i = 0; i < om.size(); ++i) om[i] = (float)(i % 9973) * 0.005f;
    CK(hipMemcpy(buf.omega, om.data(), om.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(buf.T, 0, 2 * L * pl * 32));
    std::vector<cplx> tw;
    make_twiddles(N, tw);
    cplx *twd;
    CK(hipMalloc(&twd, tw.size() * 8));
    CK(hipMemcpy(twd, tw.data(), tw.size() * 8, hipMemcpyHostToDevice));
    buf.tw = twd;
    FrameArgs args{};
    for (int i = 0; i < C; ++i) args.c[i] = CascadeFrame{88.f + i, 88.f + i, 120.5f + i, 0.5f, 0.75f, 0.9f, i, 0};
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));

    const dim3 g1(C * (N / kWgRows), 6), b1(plan_wg_threads(N));
    const dim3 g2(C * (N / plan_lp_rows(N))), b2(plan_lp_threads(N));
    const int w1 = (plan_wg_threads(N) + 63) / 64, w2 = (plan_lp_threads(N) + 63) / 64;
    Stamp *st;
    const size_t nst = std::max((size_t)g1.x * g1.y * w1, (size_t)g2.x * w2);
    CK(hipMalloc(&st, sizeof(Stamp) * nst));
    float *scratch;
    CK(hipMalloc(&scratch, 1 << 20));
    CK(hipMemset(scratch, 0, 1 << 20));

    auto p1 = [&] { hipLaunchKernelGGL((k_pass1c_lp<N>), g1, b1, 0, s, buf, args, (Stamp *)nullptr); };
    auto p2 = [&] { hipLaunchKernelGGL((k_pass2c_lp<N, false>), g2, b2, 0, s, buf, args, (Stamp *)nullptr); };
    printf("N = %d, C = %d: pass 1 grid %u x %u blocks of %u threads, pass 2 grid %u blocks of %u threads\n", N, C, g1.x, g1.y, b1.x, g2.x, b2.x);
    for (int i = 0; i < 3000; ++i) { p1(); p2(); }  // clocks
    CK(hipStreamSynchronize(s));
    const float tick = time_it([&] { p1(); p2(); }, iters, s);
    const float t1 = time_it(p1, iters, s), t2 = time_it(p2, iters, s);
    printf("tick (pass 1 + pass 2 back to back)          : %7.2f us\n", tick);
    printf("pass 1 alone, back to back                   : %7.2f us\n", t1);
    printf("pass 2 alone, back to back                   : %7.2f us\n", t2);
    printf("empty kernel, pass-1 grid                    : %7.2f us\n", time_it([&] { hipLaunchKernelGGL(k_empty, g1, b1, 0, s); }, iters, s));
    printf("empty kernel, pass-2 grid                    : %7.2f us\n", time_it([&] { hipLaunchKernelGGL(k_empty, g2, b2, 0, s); }, iters, s));
    printf("empty kernel, 1 block                        : %7.2f us\n", time_it([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); }, iters, s));
    printf("one load -> one store per wave, pass-1 grid  : %7.2f us\n", time_it([&] { hipLaunchKernelGGL(k_touch, dim3(g1.x * g1.y), b1, 0, s, scratch, scratch + (1 << 17)); }, iters, s));
    printf("one load -> one store per wave, pass-2 grid  : %7.2f us\n", time_it([&] { hipLaunchKernelGGL(k_touch, g2, b2, 0, s, scratch, scratch + (1 << 17)); }, iters, s));

    {
        CK(hipMemset(st, 0, sizeof(Stamp) * nst));
        for (int i = 0; i < 50; ++i) { p1(); p2(); }
        hipLaunchKernelGGL((k_pass1c_lp<N, kAuxDefault, true>), g1, b1, 0, s, buf, args, st);
        CK(hipStreamSynchronize(s));
        std::vector<Stamp> h((size_t)g1.x * g1.y * w1);
        CK(hipMemcpy(h.data(), st, sizeof(Stamp) * h.size(), hipMemcpyDeviceToHost));
        const char *names[15] = {"start", "loads issued", "twiddles in LDS (block barrier)", "own data arrived", "modulated", "layer input built",
                                 "transformed", "staged (block barrier)", "stores issued", "", "", "", "", "", "stores acknowledged"};
        report("k_pass1c_lp", h, names, 15, t1);
    }
    {
        CK(hipMemset(st, 0, sizeof(Stamp) * nst));
        for (int i = 0; i < 50; ++i) { p1(); p2(); }
        p1();
        hipLaunchKernelGGL((k_pass2c_lp<N, false, kAuxDefault, kAuxDefault, true>), g2, b2, 0, s, buf, args, st);
        CK(hipStreamSynchronize(s));
        std::vector<Stamp> h((size_t)g2.x * w2);
        CK(hipMemcpy(h.data(), st, sizeof(Stamp) * h.size(), hipMemcpyDeviceToHost));
        const char *names[15] = {"start", "loads issued", "twiddles in LDS (block barrier)", "own data arrived", "", "", "transformed",
                                 "all four transforms in LDS (block barrier)", "unpacked, stores issued", "", "", "", "", "", "stores acknowledged"};
        report("k_pass2c_lp", h, names, 15, t2);
    }
    {   // ---- tick groups (k_tick_group_c_lp): duration of a full launch (pass 2 of D ticks + pass 1 of the next D) and where its waves spend it ----
        //   kbench_small_<N> C iters [D [pass-1 items: 0 layer-parallel, 1 compact [pass 2: 0 plain, 1 pipelined]]]
        using TP = TickPlan<N>;
        const int D = argc > 3 ? atoi(argv[3]) : 8, p1c = argc > 4 ? atoi(argv[4]) : 0, pipe = argc > 5 ? atoi(argv[5]) : 0;
        DeviceBuffers gb = buf;
        CK(hipMalloc((void **)&gb.T, (size_t)2 * D * L * pl * 32));
        CK(hipMemset(gb.T, 0, (size_t)2 * D * L * pl * 32));
        CK(hipMalloc((void **)&gb.pcol, (size_t)2 * D * C * N * 8));
        CK(hipMalloc((void **)&gb.rrow, (size_t)2 * D * C * N * 32));
        CK(hipMemset(gb.pcol, 0, (size_t)2 * D * C * N * 8));
        CK(hipMemset(gb.rrow, 0, (size_t)2 * D * C * N * 32));
        TickGroupArgs ga{};
        ga.slots = C;
        ga.p1_compact = p1c;
        ga.p2_pipe = pipe;
        ga.d2 = ga.d1 = D;
        ga.n2 = pipe ? TP::items_2_pipe(C) : TP::items_2(C);
        ga.n1 = p1c ? TP::items_1_compact(C) : TP::items_1(C);
        int parity = 0;
        auto fill = [&] {
            for (int j = 0; j < D; ++j) {
                ga.tbase2[j] = (parity * D + j) * C;
                ga.tbase1[j] = ((parity ^ 1) * D + j) * C;
                for (int i = 0; i < C; ++i) ga.time1[j][i] = 120.5f + i + 0.02f * j;
            }
            parity ^= 1;
        };
        const int blocks = ga.n2 + D * ga.n1, wpb = plan_lp_threads(N) / 64;
        auto launch = [&](bool stamped, Stamp *out) {
            const dim3 gr(blocks), bl(plan_lp_threads(N));
            if (pipe) {
                if (stamped) hipLaunchKernelGGL((k_tick_group_c_lp<N, false, true, true>), gr, bl, 0, s, gb, args, ga, out);
                else hipLaunchKernelGGL((k_tick_group_c_lp<N, false, false, true>), gr, bl, 0, s, gb, args, ga, out);
            } else {
                if (stamped) hipLaunchKernelGGL((k_tick_group_c_lp<N, false, true, false>), gr, bl, 0, s, gb, args, ga, out);
                else hipLaunchKernelGGL((k_tick_group_c_lp<N, false, false, false>), gr, bl, 0, s, gb, args, ga, out);
            }
        };
        auto grp = [&] { fill(); launch(false, nullptr); };
        const float tg = time_it(grp, std::max(50, iters / D), s);
        printf("tick group: D = %d, pass-1 items %s, pass 2 %s: %d pass-2 blocks + %d x %d pass-1 blocks of %d threads: %7.2f us per launch = %6.2f us per tick\n", D,
               p1c ? "compact" : "layer-parallel", pipe ? "pipelined" : "plain", ga.n2, D, ga.n1, plan_lp_threads(N), tg, tg / D);
        if (argc > 6) {  // EXPERIMENT (round 6): the group's pass 2 (a serial chain through the ticks) and its pass 1 (independent of everything but scratch) as TWO
            // launches on two streams -- mode 1: event-chained (pass 2 of group k waits for pass 1 of group k, pass 1 of group k + 2 for pass 2 of group k),
            // mode 2: no dependencies at all (the bound).  Timing only.
            const int mode = atoi(argv[6]);
            hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
            constexpr int R = 8;
            hipEvent_t ea[R], eb[R];
            for (int i = 0; i < R; ++i) { CK(hipEventCreateWithFlags(&ea[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&eb[i], hipEventDisableTiming)); }
            auto go = [&](const TickGroupArgs &h, int nblocks, hipStream_t st) {
                const dim3 gr(nblocks), bl(plan_lp_threads(N));
                if (pipe) hipLaunchKernelGGL((k_tick_group_c_lp<N, false, false, true>), gr, bl, 0, st, gb, args, h, (Stamp *)nullptr);
                else hipLaunchKernelGGL((k_tick_group_c_lp<N, false, false, false>), gr, bl, 0, st, gb, args, h, (Stamp *)nullptr);
            };
            auto run = [&](int n) {
                for (int i = 0; i < n; ++i) {
                    fill();
                    TickGroupArgs h1 = ga, h2 = ga;
                    h1.d2 = 0; h1.n2 = 0;   // pass 1 of D ticks
                    h2.d1 = 0;              // pass 2 of D ticks
                    if (mode == 1 && i > 1) CK(hipStreamWaitEvent(sb, ea[(i - 2) % R], 0));
                    go(h1, D * ga.n1, sb);
                    if (mode == 1) CK(hipEventRecord(eb[i % R], sb));
                    if (mode == 1 && i > 0) CK(hipStreamWaitEvent(sa, eb[(i - 1) % R], 0));
                    go(h2, ga.n2, sa);
                    if (mode == 1) CK(hipEventRecord(ea[i % R], sa));
                }
                CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
            };
            run(20);
            for (int rep = 0; rep < 3; ++rep) {
                const int n = std::max(200, iters / D);
                auto t0 = std::chrono::steady_clock::now();
                run(n);
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                auto t1 = std::chrono::steady_clock::now();
                for (int i = 0; i < n; ++i) grp();
                CK(hipStreamSynchronize(s));
                const double us1 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count();
                printf("  two streams (pass 2 | pass 1)%s: %7.2f us per group = %6.2f per tick;   one launch per group: %7.2f = %6.2f per tick (host clock, %d groups)\n",
                       mode == 1 ? ", event-chained" : ", NO dependencies (bound)", us / n, us / n / D, us1 / n, us1 / n / D, n);
            }
            return 0;
        }
        Stamp *gs;
        CK(hipMalloc(&gs, sizeof(Stamp) * (size_t)blocks * wpb));
        CK(hipMemset(gs, 0, sizeof(Stamp) * (size_t)blocks * wpb));
        for (int i = 0; i < 20; ++i) grp();
        fill();
        launch(true, gs);
        CK(hipStreamSynchronize(s));
        std::vector<Stamp> h((size_t)blocks * wpb);
        CK(hipMemcpy(h.data(), gs, sizeof(Stamp) * h.size(), hipMemcpyDeviceToHost));
        // per-wave differences only (the counters of different XCDs are not comparable).  Pass-2 waves live through the whole launch:
        // their average life calibrates ticks per us.  Stamps: [k] = start of tick k (plain) / end of step k - 1 (pipelined), k >= 1.
        const int steps = pipe ? D + 1 : D;
        double p2_life = 0, p2_step[16] = {0}, p1_life = 0;
        int n2w = 0, n1w = 0;
        for (auto &x : h) {
            if (!x.t[0]) continue;
            if (x.t[15] < 1000) {
                ++n2w;
                p2_life += (double)(x.t[14] - x.t[0]);
                for (int j = 0; j < steps && j < 13; ++j) p2_step[j] += (double)(x.t[j + 1] - x.t[j]);
            } else {
                ++n1w;
                p1_life += (double)(x.t[14] - x.t[0]);
            }
        }
        const double per_us = p2_life / n2w / tg;
        printf("  pass-2 waves (%d) live %.0f ticks = the launch (=> %.0f ticks per us); per %s:", n2w, p2_life / n2w, per_us, pipe ? "step" : "tick");
        for (int j = 0; j < steps && j < 13; ++j) printf(" %.2f", p2_step[j] / n2w / per_us);
        printf(" us\n  pass-1 waves (%d): average life %.2f us\n", n1w, p1_life / n1w / per_us);
    }
    return 0;
}
