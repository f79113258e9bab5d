// kbench_2048pair.hip -- developer microbenchmark (not part of the product): the 2048^2 kernels one cascade per launch, as the runtime
// issues them -- k_pass1c_split + k_pass2c one launch per pass against the stream of k_tick_pair_c_split launches (pass 2 of cascade c
// beside pass 1 of cascade c + 1) -- and per-wave phase stamps of the split-plan pass 1 in both launch shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I godotoceanwaves_amd/csrc tools/kbench_2048pair.hip -o tools/kbench_2048pair
//   tools/kbench_2048pair [cascades = 4] [iterations = 40]
#include <hip/hip_runtime.h>
#include <algorithm>
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
    }
    return 0;
}
