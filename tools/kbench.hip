// kbench.hip -- developer microbenchmark (not part of the product): ablation variants of the frame
// kernels + per-wave timestamps, timed with hipEvents.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17
//   -I godotoceanwaves_amd/csrc tools/kbench.hip -o tools/kbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "ow_frame_kernels.h"
#include "ow_tables.h"

using namespace ow;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

template <class F>
float time_it(F f, int iters, hipStream_t s) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters * 1e3f;
}

int main(int argc, char **argv) {
#ifndef KBENCH_N
#define KBENCH_N 1024
#endif
    constexpr int N = KBENCH_N;
    const int C = argc > 1 ? atoi(argv[1]) : 4;
    const size_t pl = (size_t)N * N, L = C;
    DeviceBuffers buf{};
    CK(hipMalloc((void**)&buf.h0, L * pl * 8)); CK(hipMalloc(&buf.omega, L * pl * 4)); CK(hipMalloc((void**)&buf.T, L * pl * 32));
    CK(hipMalloc(&buf.disp, L * pl * 8)); CK(hipMalloc(&buf.norm, L * pl * 8)); CK(hipMalloc((void**)&buf.status, 64)); CK(hipMemset(buf.status, 0, 64)); CK(hipMalloc(&buf.foam, L * pl * 2)); CK(hipMemset(buf.foam, 0, L * pl * 2));
    const int dmode = argc > 3 ? atoi(argv[3]) : 0;  // 0 random O(1), 1 zeros, 2 spectrum-like (tiny away from the centre), 3 tiny but normal (1e-30)
    std::vector<float> hh(L * pl * 2); for (size_t i = 0; i < hh.size(); ++i) {
        float v = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f - 0.5f;
        const size_t tex = (i / 2) % pl; const int x = (int)(tex % N) - N / 2, y = (int)(tex / N) - N / 2;
        if (dmode == 1) v = 0.0f;
        if (dmode == 2) v *= (x * x + y * y < 60 * 60) ? 1.0f : 1e-42f;
        if (dmode == 3) v *= (x * x + y * y < 60 * 60) ? 1.0f : 1e-30f;
        hh[i] = v;
    }
    printf("data mode %d\n", dmode);
    CK(hipMemcpy(buf.h0, hh.data(), hh.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> om(L * pl); for (size_t i = 0; i < om.size(); ++i) om[i] = (float)(i % 9973) * 0.005f;
    CK(hipMemcpy(buf.omega, om.data(), om.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(buf.norm, 0, L * pl * 8)); CK(hipMemset(buf.T, 0, L * pl * 32));
    std::vector<cplx> tw; make_twiddles(N, tw); cplx *twd; CK(hipMalloc(&twd, tw.size() * 8)); CK(hipMemcpy(twd, tw.data(), tw.size() * 8, hipMemcpyHostToDevice)); buf.tw = twd;
    FrameArgs args{}; for (int i = 0; i < C; ++i) { args.c[i] = CascadeFrame{88.f + i, 88.f + i, 120.5f + i, 0.5f, 0.75f, 0.9f, i, 0}; }
    Stamp *st; CK(hipMalloc(&st, sizeof(Stamp) * C * N));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int blocks1 = C * (N / kWgRows), blocks2 = C * (N / kWgRows); const int thr1 = plan_wg_threads(N), thr2 = plan_wg_threads(N);
    DebugArgs dbg{st, 0};
    const int iters = argc > 2 ? atoi(argv[2]) : 30;
    for (int i = 0; i < 600; ++i) { hipLaunchKernelGGL((k_pass1<N, 0>), dim3(blocks1), dim3(thr1), 0, s, buf, args, dbg); hipLaunchKernelGGL((k_pass2<N, false, 0>), dim3(blocks2), dim3(thr2), 0, s, buf, args, dbg); }
    CK(hipStreamSynchronize(s));
#define RUN1(VAR) printf("pass1 var %2d : %8.2f us\n", VAR, time_it([&] { hipLaunchKernelGGL((k_pass1<N, VAR>), dim3(blocks1), dim3(thr1), 0, s, buf, args, dbg); }, iters, s));
#define RUN2(VAR) printf("pass2 var %2d : %8.2f us\n", VAR, time_it([&] { hipLaunchKernelGGL((k_pass2<N, false, VAR>), dim3(blocks2), dim3(thr2), 0, s, buf, args, dbg); }, iters, s));
    RUN1(0) RUN1(1) RUN1(2) RUN1(3) RUN1(4) RUN1(5) RUN1(6) RUN1(7)
    RUN2(0) RUN2(1) RUN2(2) RUN2(3) RUN2(4) RUN2(5) RUN2(6) RUN2(7)
#define RUNP1(AT, AH) printf("pass1 aux T=%d H=%d : %8.2f us\n", AT, AH, time_it([&] { hipLaunchKernelGGL((k_pass1<N, 0, AT, AH>), dim3(blocks1), dim3(thr1), 0, s, buf, args, dbg); }, iters, s));
#define RUNP2(AT, AO) printf("pass2 aux T=%d O=%d : %8.2f us\n", AT, AO, time_it([&] { hipLaunchKernelGGL((k_pass2<N, false, 0, AT, AO>), dim3(blocks2), dim3(thr2), 0, s, buf, args, dbg); }, iters, s));
#define TICK(A1T, A1H, A2T, A2O) printf("tick aux p1(T=%d H=%d) p2(T=%d O=%d) : %8.2f us\n", A1T, A1H, A2T, A2O, time_it([&] { hipLaunchKernelGGL((k_pass1<N, 0, A1T, A1H>), dim3(blocks1), dim3(thr1), 0, s, buf, args, dbg); hipLaunchKernelGGL((k_pass2<N, false, 0, A2T, A2O>), dim3(blocks2), dim3(thr2), 0, s, buf, args, dbg); }, iters, s));
    RUNP1(0, 0) RUNP1(2, 0) RUNP1(0, 2) RUNP1(2, 2) RUNP1(1, 0) RUNP1(3, 0)
    RUNP2(0, 0) RUNP2(2, 0) RUNP2(0, 2) RUNP2(2, 2) RUNP2(1, 0) RUNP2(0, 1) RUNP2(0, 3)
    TICK(0, 0, 0, 0) TICK(0, 0, 2, 2) TICK(0, 0, 0, 2) TICK(0, 2, 0, 2) TICK(0, 2, 2, 2) TICK(2, 0, 2, 2) TICK(1, 0, 1, 2)
    {   // two-stream experiment: pass 2 of tick n (reads T) next to pass 1 of tick n+1 (writes a second T), dependencies as in a real pipeline
        DeviceBuffers bufB = buf; CK(hipMalloc((void**)&bufB.T, L * pl * 32)); CK(hipMemset(bufB.T, 0, L * pl * 32));
        hipStream_t s2; CK(hipStreamCreate(&s2));
        const int K = iters; std::vector<hipEvent_t> e1(K + 2), e2(K + 2);
        for (auto &e : e1) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); for (auto &e : e2) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            CK(hipEventRecord(a, s));
            CK(hipStreamWaitEvent(s2, a, 0));
            for (int i = 0; i < K; ++i) {
                DeviceBuffers &bi = (i & 1) ? bufB : buf;
                if (i >= 2) CK(hipStreamWaitEvent(s2, e2[i - 2], 0));           // T buffer of tick i is free once pass 2 of tick i-2 has read it
                hipLaunchKernelGGL((k_pass1<N, 0>), dim3(blocks1), dim3(thr1), 0, s2, bi, args, dbg);
                CK(hipEventRecord(e1[i], s2));
                CK(hipStreamWaitEvent(s, e1[i], 0));
                hipLaunchKernelGGL((k_pass2<N, false, 0>), dim3(blocks2), dim3(thr2), 0, s, bi, args, dbg);
                CK(hipEventRecord(e2[i], s));
            }
            CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b)); CK(hipStreamSynchronize(s2));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            printf("two-stream pipelined tick : %8.2f us\n", ms / K * 1e3f);
        }
    }
    // both passes back to back (a tick)
    printf("tick        : %8.2f us\n", time_it([&] { hipLaunchKernelGGL((k_pass1<N, 0>), dim3(blocks1), dim3(thr1), 0, s, buf, args, dbg);
                                                      hipLaunchKernelGGL((k_pass2<N, false, 0>), dim3(blocks2), dim3(thr2), 0, s, buf, args, dbg); }, iters, s));
    // timestamps (shader clock cycles, s_memtime), pass 1: per-phase averages over all waves
    {
        hipLaunchKernelGGL((k_pass1<N, 8>), dim3(blocks1), dim3(thr1), 0, s, buf, args, dbg);
        CK(hipStreamSynchronize(s));
        const int waves = C * N * plan_T(N) / 64; std::vector<Stamp> h(waves); CK(hipMemcpy(h.data(), st, sizeof(Stamp) * waves, hipMemcpyDeviceToHost));
        const char *names[15] = {"start", "modulated", "L0 input", "L0 fft", "L0 stored", "L1 input", "L1 fft", "L1 stored", "L2 input", "L2 fft", "L2 stored", "L3 input", "L3 fft", "L3 stored", "drained"};
        double avg[15] = {0}; unsigned long long tmin = ~0ull; for (auto &x : h) tmin = std::min(tmin, x.t[0]);
        for (auto &x : h) for (int k = 0; k < 15; ++k) avg[k] += (double)(x.t[k] - tmin) / waves;
        printf("pass1 phase stamps (kcycles since first wave start; delta):\n");
        for (int k = 0; k < 15; ++k) printf("  %-10s %8.2f  (+%.2f)\n", names[k], avg[k] / 1e3, k ? (avg[k] - avg[k - 1]) / 1e3 : 0.0);
    }
    {   // compact pass 1: per-phase averages, upper-half and lower-half blocks apart (the kernel needs the side buffers)
        CK(hipMalloc((void**)&buf.pcol, (size_t)C * N * 8)); CK(hipMalloc((void**)&buf.rrow, (size_t)C * N * 32));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_pass1c<N, kAuxDefault, kAuxDefault, true>), dim3(blocks1), dim3(thr1), 0, s, buf, args, st);
        CK(hipStreamSynchronize(s));
        const int waves = C * N * plan_T(N) / 64; std::vector<Stamp> h(waves); CK(hipMemcpy(h.data(), st, sizeof(Stamp) * waves, hipMemcpyDeviceToHost));
        const char *names[15] = {"start", "modulated", "C0 input", "C0 fft", "C0 staged", "C1 input", "C1 fft", "C1 staged", "C2 input", "C2 fft", "C2 staged(+stores issued)", "", "", "", "drained"};
        for (int half = 0; half < 2; ++half) {  // half 0: upper blocks (rows >= N/2, three transforms), half 1: lower blocks (two)
            double avg[15] = {0}; int cnt = 0;
            for (auto &x : h) {
                const bool lower = (int)x.t[15] < N / 2;
                if ((int)lower != half) continue;
                for (int k = 0; k < 15; ++k) avg[k] += x.t[k] ? (double)(x.t[k] - x.t[0]) : 0.0;  // since this wave's own start
                ++cnt;
            }
            printf("k_pass1c phases, %s blocks (%d waves): average time since the wave's start, in ticks of the shader clock counter\n", half ? "lower (2 transforms)" : "upper (3 transforms)", cnt);
            double prev = 0;
            for (int k = 1; k < 15; ++k) if (names[k][0] && avg[k] > 0) { printf("  %-28s %9.1f  (+%.1f)\n", names[k], avg[k] / cnt, avg[k] / cnt - prev); prev = avg[k] / cnt; }
        }
    }
    return 0;
}
