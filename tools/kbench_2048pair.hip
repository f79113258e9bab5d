// kbench_2048pair.hip -- developer microbenchmark (not part of the product): the 2048^2 kernels one cascade per launch, as the runtime
// issues them -- k_pass1c_split + k_pass2c one launch per pass against the stream of k_tick_pair_c_split launches (pass 2 of cascade c
// beside pass 1 of cascade c + 1) -- and per-wave phase stamps of the split-plan pass 1 in both launch shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I godotoceanwaves_amd/csrc tools/kbench_2048pair.hip -o tools/kbench_2048pair
//   tools/kbench_2048pair [cascades = 4] [iterations = 40]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ow_frame_kernels.h"
#include "ow_tables.h"

using namespace ow;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

template <class F>
float time_it(F f, int iters, hipStream_t s) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f(i);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) f(i);
    CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters * 1e3f;
}

// Timeline of ONE launch of the pair kernel from its waves' stamps (wave 0 of every block): the cycle counters of the XCDs are not synchronised, so every
// times come from s_memrealtime (one 100 MHz counter for the device).  Kinds: pass 2, pass 1 lower half (2 transforms), pass 1 upper half (3 transforms).
static void timeline(const std::vector<Stamp> &h, int waves_per_block) {
    constexpr int N = 2048;
    unsigned long long r0 = ~0ull;
    for (size_t i = 0; i < h.size(); i += waves_per_block) if (h[i].rt0) r0 = std::min(r0, h[i].rt0);
    struct B { int kind; double st, en; unsigned cu; };
    std::vector<B> bl;
    for (size_t i = 0; i < h.size(); i += waves_per_block) {
        const Stamp &x = h[i];
        if (!x.t[0]) continue;
        const int kind = x.t[15] == 100000ull ? 0 : ((int)(x.t[15] - 1000ull) >= N / 2 ? 2 : 1);
        bl.push_back(B{kind, (double)(x.rt0 - r0) * 0.01, (double)(x.rt1 - r0) * 0.01, (x.xcc & 15) << 16 | (x.pad & 0xff00)});   // us
    }
    const char *kn[3] = {"pass 2", "pass 1, 2 transforms", "pass 1, 3 transforms"};
    auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[(size_t)(q * (v.size() - 1))]; };
    double end_max = 0; for (auto &b : bl) end_max = std::max(end_max, b.en);
    printf("timeline of one launch (%zu blocks, us since the launch's first wave by the device's 100 MHz real-time counter; last block ends at %.2f):\n", bl.size(), end_max);
    for (int gen = 0; gen < 2; ++gen)
        for (int k = 0; k < 3; ++k) {
            std::vector<double> st, en, life;
            for (auto &b : bl) if (b.kind == k && (b.st < 5.0) == (gen == 0)) { st.push_back(b.st); en.push_back(b.en); life.push_back(b.en - b.st); }
            if (st.empty()) continue;
            printf("  %s blocks of %-22s: %4zu  start p10/p50/p90 %6.2f %6.2f %6.2f   end p10/p50/p90/max %6.2f %6.2f %6.2f %6.2f   life p50 %6.2f\n", gen ? "later" : "first", kn[k], st.size(),
                   pct(st, .1), pct(st, .5), pct(st, .9), pct(en, .1), pct(en, .5), pct(en, .9), pct(en, 1.0), pct(life, .5));
        }
    const double bin = 2.0; const int nb = (int)(end_max / bin) + 1;
    printf("  blocks resident per %.0f-us bin (pass 2 / pass 1 short / pass 1 long):\n   ", bin);
    for (int i = 0; i < nb; ++i) {
        double a[3] = {0, 0, 0};
        for (auto &b : bl) { const double lo = std::max(b.st, i * bin), hi = std::min(b.en, (i + 1) * bin); if (hi > lo) a[b.kind] += (hi - lo) / bin; }
        printf(" %3.0f/%3.0f/%3.0f", a[0], a[1], a[2]);
    }
    printf("\n");
    // which kinds shared a CU in the first generation
    int both = 0, p2p2 = 0, p1p1 = 0, single = 0;
    std::vector<std::pair<unsigned, int>> first;
    for (auto &b : bl) if (b.st < 5.0) first.push_back({b.cu, b.kind ? 1 : 0});
    std::sort(first.begin(), first.end());
    for (size_t i = 0; i < first.size();) {
        size_t j = i; int n1 = 0, n2 = 0;
        while (j < first.size() && first[j].first == first[i].first) { (first[j].second ? n1 : n2)++; ++j; }
        if (n1 && n2) ++both; else if (n2 >= 2) ++p2p2; else if (n1 >= 2) ++p1p1; else ++single;
        i = j;
    }
    {   // who the stragglers are: life of first-generation blocks by the kinds that shared their CU, and ends by XCD
        std::vector<std::pair<unsigned, size_t>> byc;
        for (size_t i = 0; i < bl.size(); ++i) if (bl[i].st < 5.0) byc.push_back({bl[i].cu, i});
        std::sort(byc.begin(), byc.end());
        std::vector<double> combo[6];  // 0: P2 beside P2, 1: S beside S, 2: S beside L, 3: L beside S, 4: L beside L, 5: other
        for (size_t i = 0; i + 1 < byc.size(); i += 2) {
            if (byc[i].first != byc[i + 1].first) { --i; continue; }
            const B &a = bl[byc[i].second], &c = bl[byc[i + 1].second];
            auto cls = [](int k, int o) { return k == 0 && o == 0 ? 0 : k == 1 && o == 1 ? 1 : k == 1 && o == 2 ? 2 : k == 2 && o == 1 ? 3 : k == 2 && o == 2 ? 4 : 5; };
            combo[cls(a.kind, c.kind)].push_back(a.en - a.st);
            combo[cls(c.kind, a.kind)].push_back(c.en - c.st);
        }
        const char *cn[6] = {"pass 2 beside pass 2", "short beside short", "short beside long", "long beside short", "long beside long", "other"};
        for (int k = 0; k < 6; ++k) if (!combo[k].empty()) printf("  first generation, %-22s: %4zu blocks, life p10/p50/p90/max %6.2f %6.2f %6.2f %6.2f\n", cn[k], combo[k].size(), pct(combo[k], .1), pct(combo[k], .5), pct(combo[k], .9), pct(combo[k], 1.0));
        for (int x = 0; x < 8; ++x) {
            std::vector<double> e1, e2;
            for (auto &b : bl) if ((int)(b.cu >> 16) == x) (b.st < 5.0 ? e1 : e2).push_back(b.en);
            printf("  XCD %d: first generation ends p50/p90/max %6.2f %6.2f %6.2f   later blocks end p50/p90/max %6.2f %6.2f %6.2f\n", x, pct(e1, .5), pct(e1, .9), pct(e1, 1.0), pct(e2, .5), pct(e2, .9), pct(e2, 1.0));
        }
    }
    {   // inside a block: how far apart its waves end; between blocks: from the end of a CU's n-th block (its LAST wave) to the start of the CU's (n + 2)-th (the hand-over of a slot)
        std::vector<double> skew, gap;
        struct E { unsigned cu; double st, en_first, en_last; };
        std::vector<E> ev;
        for (size_t i = 0; i < h.size(); i += waves_per_block) {
            if (!h[i].rt0) continue;
            unsigned long long e0 = ~0ull, e1 = 0, s0 = ~0ull;
            for (int w = 0; w < waves_per_block; ++w) { e0 = std::min(e0, h[i + w].rt1); e1 = std::max(e1, h[i + w].rt1); s0 = std::min(s0, h[i + w].rt0); }
            skew.push_back((double)(e1 - e0) * 0.01);
            ev.push_back(E{(h[i].xcc & 15) << 16 | (h[i].pad & 0xff00), (double)(s0 - r0) * 0.01, (double)(e0 - r0) * 0.01, (double)(e1 - r0) * 0.01});
        }
        std::sort(ev.begin(), ev.end(), [](const E &a, const E &b) { return a.cu != b.cu ? a.cu < b.cu : a.st < b.st; });
        for (size_t i = 0; i < ev.size();) {
            size_t j = i; while (j < ev.size() && ev[j].cu == ev[i].cu) ++j;
            if (j - i == 4) {  // two first-generation blocks, two later ones: the later ones take the slots in the order the first ones free them
                std::vector<double> ends = {ev[i].en_last, ev[i + 1].en_last}; std::sort(ends.begin(), ends.end());
                gap.push_back(ev[i + 2].st - ends[0]); gap.push_back(ev[i + 3].st - ends[1]);
            }
            i = j;
        }
        printf("  inside a block, last wave's end - first wave's end: p50/p90/max %5.2f %5.2f %5.2f us;  slot hand-over (a block's last wave ends -> the next block's first wave starts): p10/p50/p90 %5.2f %5.2f %5.2f us (%zu)\n",
               pct(skew, .5), pct(skew, .9), pct(skew, 1.0), pct(gap, .1), pct(gap, .5), pct(gap, .9), gap.size());
    }
    printf("  first generation, per CU (xcc, se, sh, cu of HW_ID): one block of each pass on %d CUs, two pass-2 blocks on %d, two pass-1 blocks on %d, a single block on %d\n", both, p2p2, p1p1, single);
}

int main(int argc, char **argv) {
    constexpr int N = 2048;
    const int C = std::min(8, argc > 1 ? atoi(argv[1]) : 4), iters = argc > 2 ? atoi(argv[2]) : 40;
    const size_t pl = (size_t)N * N, L = C;
    DeviceBuffers buf{};
    CK(hipMalloc((void**)&buf.h0, L * pl * 8)); CK(hipMalloc(&buf.omega, L * pl * 4)); CK(hipMalloc((void**)&buf.T, 2 * pl * 32));
    CK(hipMalloc((void**)&buf.pcol, 2 * (size_t)N * 8)); CK(hipMalloc((void**)&buf.rrow, 2 * (size_t)N * 32));
    CK(hipMalloc(&buf.disp, L * pl * 8)); CK(hipMalloc(&buf.norm, L * pl * 8)); CK(hipMalloc((void**)&buf.status, 64)); CK(hipMemset(buf.status, 0, 64));
    CK(hipMalloc(&buf.foam, L * pl * 2)); CK(hipMemset(buf.foam, 0, L * pl * 2));
    std::vector<float> hh(L * pl * 2);
    for (size_t i = 0; i < hh.size(); ++i) hh[i] = ((float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f - 0.5f) * 1e-3f;
    CK(hipMemcpy(buf.h0, hh.data(), hh.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> om(L * pl); for (size_t i = 0; i < om.size(); ++i) om[i] = (float)(i % 9973) * 0.005f;
    CK(hipMemcpy(buf.omega, om.data(), om.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(buf.norm, 0, L * pl * 8)); CK(hipMemset(buf.T, 0, 2 * pl * 32));
    std::vector<cplx> tw, tws, twh; make_twiddles(N, tw); make_split_twiddles(N, tws); make_half_twiddles(N, twh);
    cplx *twd, *twsd, *twhd; CK(hipMalloc(&twhd, twh.size() * 8)); CK(hipMemcpy(twhd, twh.data(), twh.size() * 8, hipMemcpyHostToDevice)); CK(hipMalloc(&twd, tw.size() * 8)); CK(hipMemcpy(twd, tw.data(), tw.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&twsd, tws.size() * 8)); CK(hipMemcpy(twsd, tws.data(), tws.size() * 8, hipMemcpyHostToDevice));
    buf.tw = twd; buf.tw_split = twsd; buf.tw_half = twhd;
    FrameArgs args{}; for (int i = 0; i < C; ++i) args.c[i] = CascadeFrame{88.f + i, 88.f + i, 120.5f + i, 0.5f, 0.75f, 0.9f, i, 0};
    constexpr int W1 = SplitGeo<N, 4>::kThreads / 64, W2 = PairSplitGeo<N>::kThreads / 64, PT = PairSplitGeo<N>::kThreads;
    Stamp *st; CK(hipMalloc(&st, sizeof(Stamp) * 1024 * 16));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int b1 = (N / kWgRows) * 2, b2 = N / kWgRows;
    // one cascade per launch, as the runtime does at this size: launch slot 0 of a FrameArgs holding cascade c
    auto one = [&](int c) { FrameArgs a{}; a.c[0] = args.c[c]; return a; };
    auto p1 = [&](int i) { hipLaunchKernelGGL((k_pass1c_split<N, 4>), dim3(b1), dim3(SplitGeo<N, 4>::kThreads), 0, s, buf, one(i % C), (Stamp *)nullptr); };
    auto p2 = [&](int i) { hipLaunchKernelGGL((k_pass2c<N, false, kAuxDefault, kAuxNT>), dim3(b2), dim3(plan_wg_threads(N)), 0, s, buf, one(i % C)); };
    for (int i = 0; i < 200; ++i) { p1(i); p2(i); }
    CK(hipStreamSynchronize(s));
    printf("2048^2, %d cascades in rotation, us per CASCADE:\n", C);
    printf("  k_pass1c_split<4 rows> alone : %8.2f\n", time_it(p1, iters * C, s));
    printf("  k_pass2c alone               : %8.2f\n", time_it(p2, iters * C, s));
    printf("  one launch per pass          : %8.2f\n", time_it([&](int i) { p1(i); p2(i); }, iters * C, s));
    PairArgs g{};
    g.n2 = g.n1 = N / 4;  // 8-wave blocks of both kinds
    for (int k = 0; k < kMaxCascades; ++k) {
        const CascadeFrame &cf = args.c[k];
        g.c[k] = PairFrame{cf.tile_x, cf.tile_y, cf.whitecap, cf.foam_grow_rate, cf.foam_decay, cf.cascade};
        g.time1[k] = cf.time;
    }
    auto pair_args = [&](int i) {  // pass 2 of cascade i - 1 (scratch slot (i - 1) & 1), pass 1 of cascade i (scratch slot i & 1)
        PairArgs h = g;
        h.first2 = (i + C - 1) % C; h.first1 = i % C; h.tbase2 = (i + 1) & 1; h.tbase1 = i & 1;
        return h;
    };
    auto pair = [&](int i) { hipLaunchKernelGGL((k_tick_pair_c_split<N, false>), dim3(g.n2 + g.n1), dim3(PT), 0, s, buf, pair_args(i), (Stamp *)nullptr); };
    printf("  k_tick_pair_c_split stream   : %8.2f\n", time_it(pair, iters * C, s));
    printf("  one launch per pass (again)  : %8.2f\n", time_it([&](int i) { p1(i); p2(i); }, iters * C, s));
    printf("  k_tick_pair_c_split (again)  : %8.2f\n", time_it(pair, iters * C, s));
    if (argc > 3) {  // EXPERIMENT (round 6): pass 1 and pass 2 as SEPARATE half-filled launches of the pair kernel on two streams, each waiting for the other's
        // previous launch by an event (pass 2 of tick i needs pass 1 of tick i = B_{i-1}; pass 1 of tick i + 1 needs the scratch pass 2 of tick i - 1 read = A_{i-1}):
        // at any time one launch of each kind is in flight, a skewed pair, and the drain of one runs under the body of the other.  Timing only (the data are not chained).
        hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
        constexpr int R = 8;
        hipEvent_t ea[R], eb[R];
        for (int i = 0; i < R; ++i) { CK(hipEventCreateWithFlags(&ea[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&eb[i], hipEventDisableTiming)); }
        const bool events = atoi(argv[3]) == 1;
        auto run = [&](int n) {
            for (int i = 0; i < n; ++i) {
                PairArgs h1 = pair_args(i), h2 = pair_args(i);
                h1.n2 = 0; h2.n1 = 0;
                if (events && i > 0) CK(hipStreamWaitEvent(sb, ea[(i - 1) % R], 0));
                hipLaunchKernelGGL((k_tick_pair_c_split<N, false>), dim3(h1.n1), dim3(PT), 0, sb, buf, h1, (Stamp *)nullptr);
                if (events) CK(hipEventRecord(eb[i % R], sb));
                if (events && i > 0) CK(hipStreamWaitEvent(sa, eb[(i - 1) % R], 0));
                hipLaunchKernelGGL((k_tick_pair_c_split<N, false>), dim3(h2.n2), dim3(PT), 0, sa, buf, h2, (Stamp *)nullptr);
                if (events) CK(hipEventRecord(ea[i % R], sa));
            }
            CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
        };
        run(20);
        for (int rep = 0; rep < 3; ++rep) {
            const int n = iters * C * 4;
            const auto t0 = std::chrono::steady_clock::now();
            run(n);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf("  two streams, one launch per pass each%s: %8.2f us per cascade (host clock, %d ticks)\n", events ? ", event-chained" : ", NO dependencies (upper bound)", us / n, n);
            CK(hipStreamSynchronize(s));
            const auto t1 = std::chrono::steady_clock::now();
            for (int i = 0; i < n; ++i) pair(i);
            CK(hipStreamSynchronize(s));
            printf("  k_tick_pair_c_split stream (host clock)  : %8.2f\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count() / n);
        }
        return 0;
    }
    uint32_t status = 0; CK(hipMemcpy(&status, buf.status, 4, hipMemcpyDeviceToHost));
    printf("  status word 0x%x\n", status);

    const char *names[16] = {"start", "loads issued", "table + own data", "modulated", "C0 input", "C0 transformed", "C0 staged", "C1 input", "C1 transformed",
                             "C1 staged", "C2 input", "C2 transformed", "C2 staged", "stores issued", "stores acknowledged", ""};
    auto report = [&](const char *title, const std::vector<Stamp> &h, unsigned long long tag_lo, unsigned long long tag_hi, bool lower_only, bool upper_only) {
        double avg[16] = {0}; int cnt = 0; double life = 0;
        for (auto &x : h) {
            if (x.t[15] < tag_lo || x.t[15] >= tag_hi) continue;
            const int row0 = (int)(x.t[15] - tag_lo);
            if ((lower_only && row0 >= N / 2) || (upper_only && row0 < N / 2)) continue;
            for (int k = 0; k < 15; ++k) avg[k] += x.t[k] ? (double)(x.t[k] - x.t[0]) : 0.0;
            life += (double)(x.t[14] - x.t[0]);
            ++cnt;
        }
        if (!cnt) return;
        printf("%s (%d waves): average clocks since the wave's own start\n", title, cnt);
        double prev = 0;
        for (int k = 1; k < 15; ++k) if (avg[k] > 0) { printf("    %-22s %9.0f  (+%.0f)\n", names[k], avg[k] / cnt, avg[k] / cnt - prev); prev = avg[k] / cnt; }
    };
    {
        for (int i = 0; i < 8; ++i) hipLaunchKernelGGL((k_pass1c_split<N, 4, kAuxDefault, kAuxDefault, true>), dim3(b1), dim3(SplitGeo<N, 4>::kThreads), 0, s, buf, one(i % C), st);
        CK(hipStreamSynchronize(s));
        std::vector<Stamp> h((size_t)b1 * W1); CK(hipMemcpy(h.data(), st, sizeof(Stamp) * h.size(), hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0; for (auto &x : h) { t0 = std::min(t0, x.t[0]); t1 = std::max(t1, x.t[14]); }
        printf("k_pass1c_split alone: launch spans %llu clocks\n", t1 - t0);
        report("  upper-half items (3 transforms)", h, 0, N, false, true);
        report("  lower-half items (2 transforms)", h, 0, N, true, false);
    }
    {
        for (int i = 0; i < 8; ++i) {
            hipLaunchKernelGGL((k_tick_pair_c_split<N, false, true>), dim3(g.n2 + g.n1), dim3(PT), 0, s, buf, pair_args(i), st);
        }
        CK(hipStreamSynchronize(s));
        std::vector<Stamp> h((size_t)(g.n2 + g.n1) * W2); CK(hipMemcpy(h.data(), st, sizeof(Stamp) * h.size(), hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0; for (auto &x : h) { t0 = std::min(t0, x.t[0]); t1 = std::max(t1, x.t[14]); }
        printf("k_tick_pair_c_split: launch spans %llu clocks\n", t1 - t0);
        report("  pass-1 waves, upper-half units", h, 1000, 1000 + N, false, true);
        report("  pass-1 waves, lower-half units", h, 1000, 1000 + N, true, false);
        double life = 0, start = 0; int cnt = 0;
        for (auto &x : h) if (x.t[15] == 100000ull) { life += (double)(x.t[14] - x.t[0]); start += (double)(x.t[0] - t0); ++cnt; }
        if (cnt) printf("  pass-2 waves (%d): average life %.0f clocks, average start %.0f clocks into the launch\n", cnt, life / cnt, start / cnt);
        double s1 = 0; int c1 = 0;
        for (auto &x : h) if (x.t[15] >= 1000ull && x.t[15] < 1000ull + N) { s1 += (double)(x.t[0] - t0); ++c1; }
        if (c1) printf("  pass-1 waves (%d): average start %.0f clocks into the launch\n", c1, s1 / c1);
        timeline(h, W2);
    }
    return 0;
}
