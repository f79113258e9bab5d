// emul.cpp -- TEST INFRASTRUCTURE: steps 64 emulated lanes through the exact lane-level device code
// of godotoceanwaves_amd/csrc/ow_device.h on the CPU, phase by phase (a phase boundary is where the
// GPU kernels place wave_sync()).  Lets the index math, twiddles, layouts and orientation of the
// HIP kernels be checked against the oracle on a machine without a GPU.  Never shipped, never linked
// into libocean_waves.so.
#include <cstring>
#include <vector>

#include "ow_device.h"
#include "ow_tables.h"

using namespace ow;

namespace {

// a block of NT threads; row of thread tau = tau / T; each row has its own LDS region (plan_region_cplx)
template <int N, int NT>
struct Block {
    static constexpr int Tn = plan_T(N), P = plan_P(N);
    std::vector<cplx> lds = std::vector<cplx>(plan_region_cplx(N) * (NT / Tn));
    std::vector<cplx> tw;
    Block() { fill_twiddles<N>(tw); }
    cplx *row(int tau) { return lds.data() + (tau / Tn) * plan_region_cplx(N); }

    // row IFFT of d[tau][P] for all threads in lockstep (mirrors row_ifft<N> in ow_frame_kernels.h;
    // every loop below is one phase between two wave_sync()s)
    void row_ifft(cplx (*d)[P]) {
        for (int l = 0; l < NT; ++l) fft_stage_compute<N, 0>(d[l], l % Tn, tw.data());
        for (int l = 0; l < NT; ++l) fft_stage_write<N, 0>(d[l], l % Tn, row(l));
        for (int l = 0; l < NT; ++l) fft_stage_read<N, 1>(d[l], l % Tn, row(l));
        for (int l = 0; l < NT; ++l) fft_stage_compute<N, 1>(d[l], l % Tn, tw.data());
        if constexpr (plan_S(N) == 3) {
            for (int l = 0; l < NT; ++l) fft_stage_write<N, 1>(d[l], l % Tn, row(l));
            for (int l = 0; l < NT; ++l) fft_stage_read<N, 2>(d[l], l % Tn, row(l));
            for (int l = 0; l < NT; ++l) fft_stage_compute<N, 2>(d[l], l % Tn, tw.data());
        }
    }
};

template <int N>
void rows_fft(const float *in, float *out, int rows) {
    constexpr int Tn = plan_T(N), P = plan_P(N), RW = plan_rows_per_wave(N);
    Block<N, 64> w;
    static cplx d[64][P];
    for (int r0 = 0; r0 < rows; r0 += RW) {
        for (int l = 0; l < 64; ++l) {
            const int rw = l / Tn, t = l % Tn;
            for (int j = 0; j < P; ++j) {
                const int x = fft_in_index<N>(t, j);
                d[l][j] = cplx{in[((size_t)(r0 + rw) * N + x) * 2], in[((size_t)(r0 + rw) * N + x) * 2 + 1]};
            }
        }
        w.row_ifft(d);
        for (int l = 0; l < 64; ++l) {
            const int rw = l / Tn, t = l % Tn;
            for (int o = 0; o < P; ++o) {
                const cplx v = d[l][OutMap<N>::slot_of(o)];
                out[((size_t)(r0 + rw) * N + t + Tn * o) * 2] = v.x;
                out[((size_t)(r0 + rw) * N + t + Tn * o) * 2 + 1] = v.y;
            }
        }
    }
}

template <int N>
void frame(const float *h0, const float *omega, CascadeFrame cf, float *Tbuf, uint16_t *disp, uint16_t *norm, float *f32) {
    constexpr int Tn = plan_T(N), P = plan_P(N), RW = plan_rows_per_wave(N);
    constexpr int NT1 = 64 * plan_p1_waves(N), ROWS1 = plan_p1_rows(N);
    const float dkx = (2.0f * kPi) / cf.tile_x, dky = (2.0f * kPi) / cf.tile_y;
    f32x4 *Tc = reinterpret_cast<f32x4 *>(Tbuf);
    // ---- pass 1 (mirrors k_pass1) ----
    {
        Block<N, NT1> w;
        static cplx h[NT1][P], a[NT1][P], b[NT1][P];
        for (int row0 = 0; row0 < N; row0 += ROWS1) {
            for (int l = 0; l < NT1; ++l) {
                const int y = row0 + l / Tn;
                Pass1<N>::load_modulate(h[l], l % Tn, reinterpret_cast<const f32x4 *>(h0) + (size_t)y * N, omega + (size_t)y * N, cf.time);
            }
            for (int pair = 0; pair < 2; ++pair) {
                for (int l = 0; l < NT1; ++l) {
                    const int y = row0 + l / Tn, t = l % Tn;
                    const float ky = (float)(y - N / 2) * dky;
                    if (pair == 0) {
                        Pass1<N>::template layer_input<0>(a[l], h[l], t, ky, dkx);
                        Pass1<N>::template layer_input<1>(b[l], h[l], t, ky, dkx);
                    } else {
                        Pass1<N>::template layer_input<2>(a[l], h[l], t, ky, dkx);
                        Pass1<N>::template layer_input<3>(b[l], h[l], t, ky, dkx);
                    }
                }
                w.row_ifft(a);
                w.row_ifft(b);
                for (int r = 0; r < 2; ++r) {
                    for (int l = 0; l < NT1; ++l) Pass1<N>::stage_write(a[l], b[l], l % Tn, r, w.row(l));
                    // __syncthreads()
                    for (int l = 0; l < NT1; ++l) Pass1<N>::stage_store(l, r, pair, row0, w.lds.data(), Tc);
                    // __syncthreads()
                }
            }
        }
    }
    // ---- pass 2 (mirrors k_pass2) ----
    {
        Block<N, 64> w;
        static cplx a[64][P], b[64][P];
        static float dhy_dx[64][P];
        for (int row0 = 0; row0 < N; row0 += RW) {
            for (int l = 0; l < 64; ++l) Pass2<N>::load_pair(a[l], b[l], l % Tn, row0 + l / Tn, 0, Tc);
            w.row_ifft(a);
            w.row_ifft(b);
            for (int l = 0; l < 64; ++l) {
                const int xp = row0 + l / Tn;
                Pass2<N>::unpack_displacement(a[l], b[l], dhy_dx[l], l % Tn, xp, reinterpret_cast<u16x4 *>(disp) + (size_t)xp * N,
                                              f32 ? f32 + (size_t)xp * N * 8 : nullptr);
            }
            for (int l = 0; l < 64; ++l) Pass2<N>::load_pair(a[l], b[l], l % Tn, row0 + l / Tn, 1, Tc);
            w.row_ifft(a);
            w.row_ifft(b);
            for (int l = 0; l < 64; ++l) {
                const int xp = row0 + l / Tn;
                Pass2<N>::unpack_normal(a[l], b[l], dhy_dx[l], l % Tn, xp, cf, reinterpret_cast<u16x4 *>(norm) + (size_t)xp * N,
                                        f32 ? f32 + (size_t)xp * N * 8 : nullptr);
            }
        }
    }
}

}  // namespace

extern "C" {

int emul_rows_fft(int n, const float *in, float *out, int rows) {
    switch (n) {
        case 128: rows_fft<128>(in, out, rows); return 0;
        case 256: rows_fft<256>(in, out, rows); return 0;
        case 512: rows_fft<512>(in, out, rows); return 0;
        case 1024: rows_fft<1024>(in, out, rows); return 0;
        case 2048: rows_fft<2048>(in, out, rows); return 0;
    }
    return 1;
}

void emul_spectrum(int n, const SpectrumPC *pc, float *h0, float *omega) {
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) {
            const f32x4 v = spectrum_texel(x, y, n, *pc);
            float *o = h0 + ((size_t)y * n + x) * 4;
            o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
            omega[(size_t)y * n + x] = omega_texel(x, y, n, pc->tile_x, pc->tile_y, pc->depth);
        }
}

// one frame of one cascade: Tbuf = 4*n*n*2 floats (device layout, see t_unit), norm is read (foam) and rewritten
int emul_frame(int n, const float *h0, const float *omega, const CascadeFrame *cf, float *Tbuf, uint16_t *disp,
               uint16_t *norm, float *f32) {
    switch (n) {
        case 128: frame<128>(h0, omega, *cf, Tbuf, disp, norm, f32); return 0;
        case 256: frame<256>(h0, omega, *cf, Tbuf, disp, norm, f32); return 0;
        case 512: frame<512>(h0, omega, *cf, Tbuf, disp, norm, f32); return 0;
        case 1024: frame<1024>(h0, omega, *cf, Tbuf, disp, norm, f32); return 0;
        case 2048: frame<2048>(h0, omega, *cf, Tbuf, disp, norm, f32); return 0;
    }
    return 1;
}

void emul_sincos(int count, const float *ph, float *sn, float *cs) {
    for (int i = 0; i < count; ++i) sincos_phase(ph[i], sn[i], cs[i]);
}
}
