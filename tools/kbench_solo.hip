// kbench_solo.hip -- developer microbenchmark (not part of the product): a lone small tick as ONE launch (tools/solo_tick_experiment.h) against
// the product's two launches (k_pass1c_lp + k_pass2c_lp), timing and BITWISE comparison of the maps after K ticks from the same state.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DKS_N=256 -I godotoceanwaves_amd/csrc -I tools tools/kbench_solo.hip -o tools/kbench_solo_256
//   tools/kbench_solo_256 [cascades = 4] [iterations = 2000]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ow_frame_kernels.h"
#include "ow_tables.h"
#include "solo_tick_experiment.h"

#ifndef KS_N
#define KS_N 256
#endif
using namespace ow;
#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

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

int main(int argc, char **argv) {
    constexpr int N = KS_N;
    using TP = TickPlan<N>;
    const int C = std::min(8, argc > 1 ? atoi(argv[1]) : 4), iters = argc > 2 ? atoi(argv[2]) : 2000;
    const size_t pl = (size_t)N * N, L = C;
    DeviceBuffers buf{};
    CK(hipMalloc((void **)&buf.h0, L * pl * 8));
    CK(hipMalloc(&buf.omega, L * pl * 4));
    CK(hipMalloc((void **)&buf.T, L * pl * 32));
    CK(hipMalloc(&buf.disp, L * pl * 8));
    CK(hipMalloc(&buf.norm, L * pl * 8));
    CK(hipMalloc(&buf.foam, L * pl * 2));
    CK(hipMalloc((void **)&buf.pcol, (size_t)C * N * 8));
    CK(hipMalloc((void **)&buf.rrow, (size_t)C * N * 32));
    CK(hipMalloc((void **)&buf.status, 64));
    CK(hipMemset(buf.status, 0, 64));
    std::vector<float> hh(L * pl * 2);
    for (size_t i = 0; i < hh.size(); ++i) {
        const size_t tex = (i / 2) % pl;
        const int x = (int)(tex % N) - N / 2, y = (int)(tex / N) - N / 2;
        hh[i] = ((float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f - 0.5f) * (x * x + y * y < 40 * 40 ? 0.02f : 1e-6f);  // spectrum-like: foam stays in (0, 1)
    }
    CK(hipMemcpy(buf.h0, hh.data(), hh.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> om(L * pl);
    for (size_t i = 0; i < om.size(); ++i) om[i] = (float)(i % 9973) * 0.005f;
    CK(hipMemcpy(buf.omega, om.data(), om.size() * 4, hipMemcpyHostToDevice));
    std::vector<cplx> tw;
    make_twiddles(N, tw);
    cplx *twd;
    CK(hipMalloc(&twd, tw.size() * 8));
    CK(hipMemcpy(twd, tw.data(), tw.size() * 8, hipMemcpyHostToDevice));
    buf.tw = twd;
    FrameArgs args{};
    for (int i = 0; i < C; ++i) args.c[i] = CascadeFrame{88.f + i, 88.f + i, 120.5f + i, 0.35f, 0.75f, 0.9f, i, 0};
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned *counters, *misplaced;
    CK(hipMalloc(&counters, 64));
    CK(hipMalloc(&misplaced, 64));
    CK(hipMemset(counters, 0, 64));
    CK(hipMemset(misplaced, 0, 64));

    const dim3 g1(C * (N / kWgRows), 6), b1(plan_wg_threads(N));
    const dim3 g2(C * (N / plan_lp_rows(N))), b2(plan_lp_threads(N));
    const int n1 = TP::items_1(1), n2 = TP::items_2(1);
    unsigned done = 0;  // pass-1 arrivals per cascade so far
    float tick_time = 120.5f;
    auto set_time = [&](int k) { for (int i = 0; i < C; ++i) args.c[i].time = tick_time + i + 0.02f * k; };
    auto two = [&] {
        hipLaunchKernelGGL((k_pass1c_lp<N>), g1, b1, 0, s, buf, args, (Stamp *)nullptr);
        hipLaunchKernelGGL((k_pass2c_lp<N, false>), g2, b2, 0, s, buf, args, (Stamp *)nullptr);
    };
    auto solo = [&](bool local) {
        SoloArgs sa{};
        sa.slots = C;
        done += (unsigned)n1;
        for (int i = 0; i < 8; ++i) sa.target[i] = done;
        if (local) hipLaunchKernelGGL((k_tick_solo_c_lp<N, false, true>), dim3(8 * (n1 + n2)), b2, 0, s, buf, args, sa, counters, misplaced);
        else hipLaunchKernelGGL((k_tick_solo_c_lp<N, false, false>), dim3(C * (n1 + n2)), b2, 0, s, buf, args, sa, counters, misplaced);
    };
    printf("N = %d, C = %d: per cascade %d pass-1 blocks + %d pass-2 blocks of %d threads\n", N, C, n1, n2, plan_lp_threads(N));
    // ---- correctness: K ticks from the same state, maps compared bit for bit with the two-launch path ----
    const int K = 200;
    std::vector<uint16_t> ref(L * pl * 8), got(L * pl * 8);
    auto run = [&](int mode) {
        CK(hipMemsetAsync(buf.foam, 0, L * pl * 2, s));
        CK(hipMemsetAsync(buf.norm, 0, L * pl * 8, s));
        for (int k = 0; k < K; ++k) {
            set_time(k);
            if (mode == 0) two();
            else solo(mode == 2);
        }
        CK(hipStreamSynchronize(s));
    };
    auto fetch = [&](std::vector<uint16_t> &v) {
        CK(hipMemcpy(v.data(), buf.disp, L * pl * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(v.data() + L * pl * 4, buf.norm, L * pl * 8, hipMemcpyDeviceToHost));
    };
    run(0);
    fetch(ref);
    for (int mode = 1; mode <= 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            run(mode);
            fetch(got);
            size_t bad = 0;
            for (size_t i = 0; i < ref.size(); ++i) bad += ref[i] != got[i];
            unsigned mis = 0, st = 0;
            CK(hipMemcpy(&mis, misplaced, 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&st, buf.status, 4, hipMemcpyDeviceToHost));
            printf("%s, %d ticks: %zu of %zu map words differ from the two-launch path; misplaced blocks so far %u; status 0x%x\n",
                   mode == 1 ? "one launch, any placement (sc1 stores + sc1 loads)" : "one launch, XCD-local (plain stores + sc1 loads) ", K, bad, ref.size(), mis, st);
        }
    }
    // ---- timing: lone ticks back to back ----
    for (int i = 0; i < 3000; ++i) two();
    CK(hipStreamSynchronize(s));
    for (int rep = 0; rep < 2; ++rep) {
        printf("two launches per tick (k_pass1c_lp + k_pass2c_lp)  : %7.2f us per tick\n", time_it(two, iters, s));
        printf("one launch, any placement (sc1 stores + sc1 loads)  : %7.2f us per tick\n", time_it([&] { solo(false); }, iters, s));
        printf("one launch, XCD-local (plain stores + sc1 loads)    : %7.2f us per tick\n", time_it([&] { solo(true); }, iters, s));
    }
    unsigned mis = 0, st = 0;
    CK(hipMemcpy(&mis, misplaced, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&st, buf.status, 4, hipMemcpyDeviceToHost));
    printf("misplaced blocks in all XCD-local launches: %u; status word 0x%x\n", mis, st);
    return 0;
}
