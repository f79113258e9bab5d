// ow_tables.h -- host-side construction of the constant twiddle table (replaces the reference's
// fft_butterfly.glsl dispatch, wave_generator.gd:52-54: the Stockham read indices are closed-form in
// this design, only exp(+2*pi*i*p*k/n_j) per radix pass is tabulated, in FP64 rounded once to FP32).
#pragma once
#include <cmath>
#include <vector>

#include "ow_device.h"

namespace ow {

template <int N>
inline void fill_twiddles(std::vector<cplx> &tw) {
    tw.assign(plan_tw_total(N) > 0 ? plan_tw_total(N) : 1, cplx{1.0f, 0.0f});
    constexpr int S = plan_S(N);
    for (int j = 0; j + 1 < S; ++j) {
        const int R = plan_R(N, j), m = plan_m(N, j), n = plan_n(N, j), off = plan_tw_off(N, j);
        for (int k = 1; k < R; ++k)
            for (int p = 0; p < m; ++p) {
                const double a = 2.0 * 3.14159265358979323846 * (double)((long long)p * k % n) / (double)n;
                tw[off + (k - 1) * m + p] = cplx{(float)std::cos(a), (float)std::sin(a)};
            }
    }
}

// split plan (N = 2048): [table of the N/2 plan][W_N^k = exp(+2 pi i k / N), k = 0 .. N/2 - 1], FP64 rounded once to FP32
template <int N>
inline void fill_split_twiddles(std::vector<cplx> &tw) {
    fill_twiddles<N / 2>(tw);
    tw.resize(plan_tw_total(N / 2));
    for (int k = 0; k < N / 2; ++k) {
        const double a = 2.0 * 3.14159265358979323846 * (double)k / (double)N;
        tw.push_back(cplx{(float)std::cos(a), (float)std::sin(a)});
    }
}
inline bool make_split_twiddles(int n, std::vector<cplx> &tw) {
    if (n == 2048) {
        fill_split_twiddles<2048>(tw);
        return true;
    }
    tw.clear();
    return false;
}

// half table (ow_device.h, fft_stage_compute<N, J, TWH>): [15][64] exp(+2 pi i p k / N) for the first wave's lanes p < 64, then the stage-1
// block of the full table unchanged; FP64 rounded once to FP32
template <int N>
inline void fill_half_twiddles(std::vector<cplx> &tw) {
    std::vector<cplx> full;
    fill_twiddles<N>(full);
    tw.assign(plan_twh_total(N), cplx{1.0f, 0.0f});
    const int R = plan_R(N, 0), m = plan_m(N, 0);
    for (int k = 1; k < R; ++k)
        for (int p = 0; p < kTwhCols; ++p) tw[(k - 1) * kTwhCols + p] = full[(k - 1) * m + p];
    for (int i = 0; i < plan_tw_size(N, 1); ++i) tw[plan_twh_off(N, 1) + i] = full[plan_tw_off(N, 1) + i];
}
inline bool make_half_twiddles(int n, std::vector<cplx> &tw) {
    if (n == 2048) {
        fill_half_twiddles<2048>(tw);
        return true;
    }
    tw.clear();
    return false;
}

inline bool make_twiddles(int n, std::vector<cplx> &tw) {
    switch (n) {
        case 128: fill_twiddles<128>(tw); return true;
        case 256: fill_twiddles<256>(tw); return true;
        case 512: fill_twiddles<512>(tw); return true;
        case 1024: fill_twiddles<1024>(tw); return true;
        case 2048: fill_twiddles<2048>(tw); return true;
    }
    return false;
}

}  // namespace ow
