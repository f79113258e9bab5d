// emul.cpp -- TEST INFRASTRUCTURE: steps 64 emulated lanes through the exact lane-level device code
// of godotoceanwaves_amd/csrc/ow_device.h on the CPU, phase by phase (a phase boundary is where the
// GPU kernels place wave_sync()).  Lets the index math, twiddles, layouts and orientation of the
// HIP kernels be checked against the oracle on a machine without a GPU.  Never shipped, never linked
// into libocean_waves.so.
#include <cstring>
#include <utility>
#include <vector>

#include "ow_device.h"
#include "ow_tables.h"

using namespace ow;

namespace {

// a block of NT threads; row of thread tau = tau / T; each row has its own LDS region (plan_region_cplx)
template <int N, int NT>
struct Block {
    static constexpr int Tn = plan_T(N), P = kP;
    std::vector<cplx> lds = std::vector<cplx>(plan_region_cplx(N) * (NT / Tn));
    std::vector<cplx> tw, twh;
    bool half_table = false;  // N = 2048: the compact pass 2's half twiddle table (ow_device.h "HALF TABLE")
    Block() {
        fill_twiddles<N>(tw);
        if constexpr (plan_row_spans_waves(N)) fill_half_twiddles<N>(twh);
    }
    cplx *row(int tau) { return lds.data() + (tau / Tn) * plan_region_cplx(N); }

    // row IFFT of d[tau][P] for all threads in lockstep (mirrors row_ifft<N> in ow_frame_kernels.h;
    // every loop below is one phase between two row_sync()s)
    void row_ifft(cplx (*d)[P]) {
        if constexpr (plan_row_spans_waves(N)) {
            if (half_table) {
                for (int l = 0; l < NT; ++l) fft_stage_compute<N, 0, true>(d[l], l % Tn, twh.data());
                for (int l = 0; l < NT; ++l) fft_stage_write<N, 0>(d[l], l % Tn, row(l));
                for (int l = 0; l < NT; ++l) fft_stage_read<N, 1>(d[l], l % Tn, row(l));
                for (int l = 0; l < NT; ++l) fft_stage_compute<N, 1, true>(d[l], l % Tn, twh.data());
                for (int l = 0; l < NT; ++l) fft_stage_write<N, 1>(d[l], l % Tn, row(l));
                for (int l = 0; l < NT; ++l) fft_stage_read<N, 2>(d[l], l % Tn, row(l));
                for (int l = 0; l < NT; ++l) fft_stage_compute<N, 2>(d[l], l % Tn, tw.data());
                return;
            }
        }
        for (int l = 0; l < NT; ++l) fft_stage_compute<N, 0>(d[l], l % Tn, tw.data());
        for (int l = 0; l < NT; ++l) fft_stage_write<N, 0>(d[l], l % Tn, row(l));
        for (int l = 0; l < NT; ++l) fft_stage_read<N, 1>(d[l], l % Tn, row(l));
        for (int l = 0; l < NT; ++l) fft_stage_compute<N, 1>(d[l], l % Tn, tw.data());
        if constexpr (plan_S(N) == 3) {
            if constexpr (plan_lane_exchange(N)) {
                // 64-lane emulation of fft_lane_exchange: the same swap plan, wave by wave
                for (int w0 = 0; w0 < NT; w0 += 64) {
                    cplx(*w)[P] = d + w0;
                    cplx o[64][P];
                    lane_exchange_plan<N>(
                        [&](int a, int b) {  // v_permlane32_swap: lanes 32..63 of A <-> lanes 0..31 of B
                            for (int l = 0; l < 32; ++l) std::swap(w[32 + l][a], w[l][b]);
                        },
                        [&](int a, int b) {  // v_permlane16_swap: odd 16-lane rows of A <-> even rows of B
                            for (int rw = 0; rw < 2; ++rw)
                                for (int l = 0; l < 16; ++l) std::swap(w[32 * rw + 16 + l][a], w[32 * rw + l][b]);
                        },
                        [&](int dst, int src) {
                            for (int l = 0; l < 64; ++l) o[l][dst] = w[l][src];
                        });
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < P; ++j) w[l][j] = o[l][j];
                }
            } else {
                for (int l = 0; l < NT; ++l) fft_stage_write<N, 1>(d[l], l % Tn, row(l));
                for (int l = 0; l < NT; ++l) fft_stage_read<N, 2>(d[l], l % Tn, row(l));
            }
            for (int l = 0; l < NT; ++l) fft_stage_compute<N, 2>(d[l], l % Tn, tw.data());
        }
    }
};

template <int N>
void rows_fft(const float *in, float *out, int rows, bool half_table = false) {
    constexpr int Tn = plan_T(N), P = kP, NT = plan_wg_threads(N);
    Block<N, NT> w;
    w.half_table = half_table;
    static cplx d[NT][P];
    for (int r0 = 0; r0 < rows; r0 += kWgRows) {
        for (int l = 0; l < NT; ++l) {
            const int rw = l / Tn, t = l % Tn;
            for (int j = 0; j < P; ++j) {
                const int x = fft_in_index<N>(t, j);
                d[l][j] = cplx{in[((size_t)(r0 + rw) * N + x) * 2], in[((size_t)(r0 + rw) * N + x) * 2 + 1]};
            }
        }
        w.row_ifft(d);
        for (int l = 0; l < NT; ++l) {
            const int rw = l / Tn, t = l % Tn;
            for (int o = 0; o < P; ++o) {
                const cplx v = d[l][OutMap<N>::slot_of(o)];
                out[((size_t)(r0 + rw) * N + t + Tn * o) * 2] = v.x;
                out[((size_t)(r0 + rw) * N + t + Tn * o) * 2 + 1] = v.y;
            }
        }
    }
}

// Pass 2 of the layer-parallel kernels (k_pass2_lp / k_pass2c_lp): the four transforms of a row are done by four lane groups, left
// in natural order, and every texel is finished by Pass2::unpack_texel from the four values.  `compact`: the groups compute
// F0..F3 from the compact intermediate and the combine step maps them back to the reference's packing.  Pass 1 must have
// run before (frame<N> or frame_compact<N> up to the end of its pass 1): here T / the side buffers are inputs.
template <int N>
void pass2_lp(bool compact, CascadeFrame cf, const float *Tbuf, const cplx *pcol, const cplx *rrow, uint16_t *disp, uint16_t *norm, uint16_t *foam, float *f32) {
    constexpr int Tn = plan_T(N), P = kP, NT = plan_wg_threads(N);
    const float dky = (2.0f * kPi) / cf.tile_y;
    const uint32_t plane = (uint32_t)N * N;
    const GBuf T_c = make_gbuf(Tbuf, t_cascade_bytes(N)), pcol_c = make_gbuf(pcol, (uint32_t)N * 8u), rrow_c = make_gbuf(rrow, (uint32_t)N * 32u);
    const GBuf disp_c = make_gbuf(disp, plane * 8u), norm_c = make_gbuf(norm, plane * 8u), foam_c = make_gbuf(foam, plane * 2u), f32_c = make_gbuf(f32, plane * 32u);
    Block<N, NT> w;
    static cplx f[4][NT][P];
    std::vector<cplx> nat((size_t)4 * NT * P);  // [group][thread][o]: the LDS regions in natural order
    for (int row0 = 0; row0 < N; row0 += kWgRows) {
        auto xp_of = [&](int l) { return row0 + l / Tn; };
        for (int g = 0; g < 4; ++g) {
            for (int l = 0; l < NT; ++l) {
                const int t = l % Tn, xp = xp_of(l);
                if (!compact) {
                    Pass2<N>::template load_layer<0>(f[g][l], t, xp, g, T_c);
                } else if constexpr (Tn >= 16) {
                    if (g == 0) Pass2<N>::template load_layer<0>(f[g][l], t, xp, 0, T_c);
                    if (g == 1) { Pass2<N>::template load_layer<0>(f[g][l], t, xp, 0, T_c); Pass2<N>::derive_dx(f[g][l], t, xp, dky, pcol_c); }
                    if (g == 2) Pass2<N>::template load_c1<0>(f[g][l], t, xp, dky, T_c);
                    if (g == 3) Pass2<N>::template load_layer<0>(f[g][l], t, xp, 2, T_c);
                    if (g != 0) Pass2<N>::put_row0(f[g][l], t, gload8(rrow_c, (uint32_t)xp * 32u, (uint32_t)g * 8u));
                }
            }
            w.row_ifft(f[g]);
            for (int l = 0; l < NT; ++l)
                for (int o = 0; o < P; ++o) nat[((size_t)g * NT + l) * P + o] = f[g][l][OutMap<N>::slot_of(o)];
        }
        for (int l = 0; l < NT; ++l) {
            const int t = l % Tn, xp = xp_of(l);
            const uint32_t tex = (uint32_t)(xp * N + t);
            for (int o = 0; o < P; ++o) {
                const cplx a = nat[((size_t)0 * NT + l) * P + o], b = nat[((size_t)1 * NT + l) * P + o], c = nat[((size_t)2 * NT + l) * P + o], d = nat[((size_t)3 * NT + l) * P + o];
                cplx l0 = a, l1 = b, l2 = c, l3 = d;
                if (compact) { l1 = cplx{c.x, b.y}; l2 = cplx{d.x, b.x}; l3 = cplx{d.y, c.y}; }
                uint16_t *fp = foam + Pass2<N>::foam_index(xp, t) + o;
                const uint16_t prev = *fp;
                *fp = f32 ? Pass2<N>::template unpack_texel<true, 0>(l0, l1, l2, l3, prev, t, xp, o, tex, cf, disp_c, norm_c, f32_c)
                          : Pass2<N>::template unpack_texel<false, 0>(l0, l1, l2, l3, prev, t, xp, o, tex, cf, disp_c, norm_c, f32_c);
            }
        }
    }
}

template <int N>
void frame(const float *h0a, const float *omega, CascadeFrame cf, float *Tbuf, uint16_t *disp, uint16_t *norm, uint16_t *foam, float *f32, bool lp = false) {
    constexpr int Tn = plan_T(N), P = kP, NT = plan_wg_threads(N);
    const float dkx = (2.0f * kPi) / cf.tile_x, dky = (2.0f * kPi) / cf.tile_y;
    const uint32_t plane = (uint32_t)N * N;
    const GBuf h0_c = make_gbuf(h0a, plane * 8u), om_c = make_gbuf(omega, plane * 4u), T_c = make_gbuf(Tbuf, t_cascade_bytes(N));
    const GBuf disp_c = make_gbuf(disp, plane * 8u), norm_c = make_gbuf(norm, plane * 8u), foam_c = make_gbuf(foam, plane * 2u), f32_c = make_gbuf(f32, plane * 32u);
    Block<N, NT> w;
    // ---- pass 1 (mirrors k_pass1) ----
    {
        static cplx h[NT][P], d[NT][P];
        static float ik[NT][P];
        for (int row0 = 0; row0 < N; row0 += kWgRows) {
            for (int l = 0; l < NT; ++l) {
                const int y = row0 + l / Tn, t = l % Tn;
                Pass1<N>::load_modulate(h[l], t, y, h0_c, om_c, cf.time);
            }
            for (int l = 0; l < NT; ++l) Pass1<N>::wave_numbers(ik[l], l % Tn, (float)(row0 + l / Tn - N / 2) * dky, dkx);
            for (int L = 0; L < kLayers; ++L) {
                for (int l = 0; l < NT; ++l) {
                    const int y = row0 + l / Tn, t = l % Tn;
                    const float ky = (float)(y - N / 2) * dky;
                    if (L == 0) Pass1<N>::template layer_input<0>(d[l], h[l], ik[l], t, ky, dkx);
                    if (L == 1) Pass1<N>::template layer_input<1>(d[l], h[l], ik[l], t, ky, dkx);
                    if (L == 2) Pass1<N>::template layer_input<2>(d[l], h[l], ik[l], t, ky, dkx);
                    if (L == 3) Pass1<N>::template layer_input<3>(d[l], h[l], ik[l], t, ky, dkx);
                }
                w.row_ifft(d);
                for (int l = 0; l < NT; ++l) Pass1<N>::stage_write(d[l], l % Tn, w.row(l));
                // lds_barrier()
                for (int l = 0; l < NT; ++l) Pass1<N>::template stage_store<0>(l, L, row0, w.lds.data(), T_c);
                // lds_barrier()
            }
        }
    }
    if (lp) {  // k_pass1_lp computes the same rows with the same lane code, one layer per block: T is what pass 1 above left
        pass2_lp<N>(false, cf, Tbuf, nullptr, nullptr, disp, norm, foam, f32);
        return;
    }
    // ---- pass 2 (mirrors k_pass2: layers 2, 3, 1, 0) ----
    {
        static cplx l2[NT][P], l3[NT][P], l1[NT][P], l0[NT][P];
        static float dhx_dx[NT][P], hz[NT][P];
        static uint32_t gy_foam[NT][P];
        static uint32_t foam_pk[NT][P / 2];
        for (int row0 = 0; row0 < N; row0 += kWgRows) {
            auto xp_of = [&](int l) { return row0 + l / Tn; };
            auto tex_of = [&](int l) { return (uint32_t)(xp_of(l) * N + l % Tn); };
            for (int l = 0; l < NT; ++l) Pass2<N>::template load_layer<0>(l2[l], l % Tn, xp_of(l), 2, T_c);
            w.row_ifft(l2);
            for (int l = 0; l < NT; ++l) {
                Pass2<N>::template load_layer<0>(l3[l], l % Tn, xp_of(l), 3, T_c);
                Pass2<N>::load_foam(foam_pk[l], l % Tn, xp_of(l), foam_c);
            }
            w.row_ifft(l3);
            for (int l = 0; l < NT; ++l) {
                if (f32) Pass2<N>::template after_layer3<true>(l3[l], l2[l], foam_pk[l], gy_foam[l], tex_of(l), cf, f32_c);
                else Pass2<N>::template after_layer3<false>(l3[l], l2[l], foam_pk[l], gy_foam[l], tex_of(l), cf, f32_c);
                Pass2<N>::store_foam(foam_pk[l], l % Tn, xp_of(l), foam_c);
                for (int o = 0; o < P; ++o) dhx_dx[l][o] = l2[l][OutMap<N>::slot_of(o)].y;
            }
            for (int l = 0; l < NT; ++l) Pass2<N>::template load_layer<0>(l1[l], l % Tn, xp_of(l), 1, T_c);
            w.row_ifft(l1);
            for (int l = 0; l < NT; ++l) {
                if (f32) Pass2<N>::template after_layer1<true, 0>(l1[l], dhx_dx[l], gy_foam[l], tex_of(l), norm_c, f32_c);
                else Pass2<N>::template after_layer1<false, 0>(l1[l], dhx_dx[l], gy_foam[l], tex_of(l), norm_c, f32_c);
                for (int o = 0; o < P; ++o) hz[l][o] = l1[l][OutMap<N>::slot_of(o)].x;
            }
            for (int l = 0; l < NT; ++l) Pass2<N>::template load_layer<0>(l0[l], l % Tn, xp_of(l), 0, T_c);
            w.row_ifft(l0);
            for (int l = 0; l < NT; ++l) {
                if (f32) Pass2<N>::template after_layer0<true, 0>(l0[l], hz[l], l % Tn, xp_of(l), tex_of(l), disp_c, f32_c);
                else Pass2<N>::template after_layer0<false, 0>(l0[l], hz[l], l % Tn, xp_of(l), tex_of(l), disp_c, f32_c);
            }
        }
    }
}

// the same frame through the compact (three-layer) intermediate: mirrors k_pass1c / k_pass2c
template <int N>
void frame_compact(const float *h0a, const float *omega, CascadeFrame cf, float *Tbuf, uint16_t *disp, uint16_t *norm, uint16_t *foam, float *f32, bool lp = false) {
    constexpr int Tn = plan_T(N), P = kP, NT = plan_wg_threads(N);
    const float dkx = (2.0f * kPi) / cf.tile_x, dky = (2.0f * kPi) / cf.tile_y;
    const uint32_t plane = (uint32_t)N * N;
    const GBuf h0_c = make_gbuf(h0a, plane * 8u), om_c = make_gbuf(omega, plane * 4u), T_c = make_gbuf(Tbuf, t_cascade_bytes(N));
    const GBuf disp_c = make_gbuf(disp, plane * 8u), norm_c = make_gbuf(norm, plane * 8u), foam_c = make_gbuf(foam, plane * 2u), f32_c = make_gbuf(f32, plane * 32u);
    std::vector<cplx> pcol(N), rrow((size_t)N * 4);
    const GBuf pcol_c = make_gbuf(pcol.data(), (uint32_t)N * 8u), rrow_c = make_gbuf(rrow.data(), (uint32_t)N * 32u);
    Block<N, NT> w;
    {   // ---- pass 1 ----
        static cplx h[NT][P], d[NT][P];
        static float ik[NT][P];
        for (int row0 = 0; row0 < N; row0 += kWgRows) {
            for (int l = 0; l < NT; ++l) Pass1<N>::load_modulate(h[l], l % Tn, row0 + l / Tn, h0_c, om_c, cf.time);
            for (int l = 0; l < NT; ++l) Pass1<N>::wave_numbers(ik[l], l % Tn, (float)(row0 + l / Tn - N / 2) * dky, dkx);
            for (int l = 0; l < NT; ++l)
                if (l % Tn == 0) gstore8(pcol_c, Pass2<N>::pcol_index(row0 + l / Tn) * 8u, 0u, Pass1<N>::column_term(h[l], ik[l], 0, dkx));
            if (row0 == 0) {  // texel row 0: three extra transforms, results to the side buffer (only the lanes of row 0 matter)
                for (int Q = 1; Q <= 3; ++Q) {
                    for (int l = 0; l < NT; ++l) {
                        const float ky = (float)(l / Tn - N / 2) * dky;
                        if (Q == 1) Pass1<N>::template row0_input<1>(d[l], h[l], ik[l], l % Tn, ky, dkx);
                        if (Q == 2) Pass1<N>::template row0_input<2>(d[l], h[l], ik[l], l % Tn, ky, dkx);
                        if (Q == 3) Pass1<N>::template row0_input<3>(d[l], h[l], ik[l], l % Tn, ky, dkx);
                    }
                    w.row_ifft(d);
                    for (int l = 0; l < Tn; ++l)
                        for (int o = 0; o < P; ++o) gstore8(rrow_c, (uint32_t)(l + Tn * o) * 32u, (uint32_t)Q * 8u, d[l][OutMap<N>::slot_of(o)]);
                }
            }
            for (int L = 0; L < Pass1<N>::kCompactLayers; ++L) {
                if (L == 1 && row0 < N / 2) continue;  // hz of the lower rows is the conjugate of the mirrored rows': not transformed
                for (int l = 0; l < NT; ++l) {
                    const int y = row0 + l / Tn, t = l % Tn;
                    const float ky = (float)(y - N / 2) * dky;
                    if (L == 0) Pass1<N>::template layer_input_c<0>(d[l], h[l], ik[l], t, ky, dkx);
                    if (L == 1) Pass1<N>::template layer_input_c<1>(d[l], h[l], ik[l], t, ky, dkx);
                    if (L == 2) Pass1<N>::template layer_input_c<2>(d[l], h[l], ik[l], t, ky, dkx);
                }
                w.row_ifft(d);
                for (int l = 0; l < NT; ++l) Pass1<N>::stage_write(d[l], l % Tn, w.row(l));
                for (int l = 0; l < NT; ++l) Pass1<N>::template stage_store<0>(l, L, row0, w.lds.data(), T_c);
            }
        }
    }
    if (lp) {
        pass2_lp<N>(true, cf, Tbuf, pcol.data(), rrow.data(), disp, norm, foam, f32);
        return;
    }
    {   // ---- pass 2: F2, F0, F1, F3 ----
        static cplx f[NT][P], c0[NT][P];
        static float c2[NT][P], dxx[NT][P];
        static uint32_t hz_pk[NT][P / 2], gx_pk[NT][P / 2], foam_pk[NT][P / 2];
        for (int row0 = 0; row0 < N; row0 += kWgRows) {
            auto xp_of = [&](int l) { return row0 + l / Tn; };
            auto tex_of = [&](int l) { return (uint32_t)(xp_of(l) * N + l % Tn); };
            auto r_of = [&](int l, int q) { return gload8(rrow_c, (uint32_t)xp_of(l) * 32u, (uint32_t)q * 8u); };
            for (int l = 0; l < NT; ++l) {
                Pass2<N>::template load_c1<0>(f[l], l % Tn, xp_of(l), dky, T_c);
                Pass2<N>::put_row0(f[l], l % Tn, r_of(l, 2));
            }
            w.row_ifft(f);
            for (int l = 0; l < NT; ++l) {
                if (f32) Pass2<N>::template after_f2<true>(f[l], hz_pk[l], c2[l], tex_of(l), f32_c);
                else Pass2<N>::template after_f2<false>(f[l], hz_pk[l], c2[l], tex_of(l), f32_c);
            }
            for (int l = 0; l < NT; ++l) {
                Pass2<N>::template load_layer<0>(c0[l], l % Tn, xp_of(l), 0, T_c);
                for (int j = 0; j < P; ++j) f[l][j] = c0[l][j];
            }
            w.row_ifft(f);
            for (int l = 0; l < NT; ++l) {
                if (f32) Pass2<N>::template after_f0<true, 0>(f[l], hz_pk[l], l % Tn, xp_of(l), tex_of(l), disp_c, f32_c);
                else Pass2<N>::template after_f0<false, 0>(f[l], hz_pk[l], l % Tn, xp_of(l), tex_of(l), disp_c, f32_c);
            }
            for (int l = 0; l < NT; ++l) {
                Pass2<N>::derive_dx(c0[l], l % Tn, xp_of(l), dky, pcol_c);
                Pass2<N>::put_row0(c0[l], l % Tn, r_of(l, 1));
            }
            w.row_ifft(c0);
            for (int l = 0; l < NT; ++l) {
                if (f32) Pass2<N>::template after_f1<true>(c0[l], dxx[l], gx_pk[l], tex_of(l), f32_c);
                else Pass2<N>::template after_f1<false>(c0[l], dxx[l], gx_pk[l], tex_of(l), f32_c);
            }
            for (int l = 0; l < NT; ++l) {
                Pass2<N>::template load_layer<0>(f[l], l % Tn, xp_of(l), 2, T_c);
                Pass2<N>::put_row0(f[l], l % Tn, r_of(l, 3));
                Pass2<N>::load_foam(foam_pk[l], l % Tn, xp_of(l), foam_c);
            }
            w.row_ifft(f);
            for (int l = 0; l < NT; ++l) {
                if (f32) Pass2<N>::template after_f3<true, 0>(f[l], dxx[l], c2[l], gx_pk[l], foam_pk[l], tex_of(l), cf, norm_c, f32_c);
                else Pass2<N>::template after_f3<false, 0>(f[l], dxx[l], c2[l], gx_pk[l], foam_pk[l], tex_of(l), cf, norm_c, f32_c);
                Pass2<N>::store_foam(foam_pk[l], l % Tn, xp_of(l), foam_c);
            }
        }
    }
}

}  // namespace

// every (L, slot, row0) the tick-group kernel's pass-1 item decode hands out for `slots` cascades: 3 ints per entry, returns the count
template <int N>
static int tick_items(int slots, int *out) {
    using TP = TickPlan<N>;
    int cnt = 0;
    for (int item = 0; item < TP::items_1(slots); ++item)
        for (int sub = 0; sub < TP::Q; ++sub) {
            int L, slot, row0;
            if (!TP::decode(item, sub, slots, L, slot, row0)) continue;
            out[3 * cnt] = L;
            out[3 * cnt + 1] = slot;
            out[3 * cnt + 2] = row0;
            ++cnt;
        }
    return cnt;
}

// the same for the k_pass1c-shaped items (TickGroupArgs::p1_compact): 3 ints per entry = (block, slot, row0)
template <int N>
static int tick_items_compact(int slots, int *out) {
    using TP = TickPlan<N>;
    int cnt = 0;
    for (int item = 0; item < TP::items_1_compact(slots); ++item)
        for (int sub = 0; sub < TP::Q; ++sub) {
            int slot, row0;
            TP::decode_compact(item, sub, slot, row0);
            out[3 * cnt] = item;
            out[3 * cnt + 1] = slot;
            out[3 * cnt + 2] = row0;
            ++cnt;
        }
    return cnt;
}

extern "C" {
int emul_tick_items_compact(int n, int slots, int *out) {
    switch (n) {
        case 256: return tick_items_compact<256>(slots, out);
        case 512: return tick_items_compact<512>(slots, out);
        case 1024: return tick_items_compact<1024>(slots, out);
    }
    return -1;
}
int emul_tick_items(int n, int slots, int *out) {
    switch (n) {
        case 256: return tick_items<256>(slots, out);
        case 512: return tick_items<512>(slots, out);
        case 1024: return tick_items<1024>(slots, out);
    }
    return -1;
}
int emul_tick_items_2(int n, int slots) { return n == 256 ? TickPlan<256>::items_2(slots) : n == 512 ? TickPlan<512>::items_2(slots) : TickPlan<1024>::items_2(slots); }


// the 2048-point row transform with the compact pass 2's half twiddle table
int emul_rows_fft_half_table(const float *in, float *out, int rows) {
    rows_fft<2048>(in, out, rows, true);
    return 0;
}

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

// h0 = the reference's float4 texels (for the comparison with the oracle), h0a = the plane the device stores (.xy only)
void emul_spectrum(int n, const SpectrumPC *pc, float *h0, float *h0a, float *omega) {
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) {
            const f32x4 v = spectrum_texel(x, y, n, *pc);
            float *o = h0 + ((size_t)y * n + x) * 4;
            o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
            const cplx a = spectrum_amplitude(x, y, n, *pc);
            h0a[((size_t)y * n + x) * 2] = a.x;
            h0a[((size_t)y * n + x) * 2 + 1] = a.y;
            omega[(size_t)y * n + x] = omega_texel(x, y, n, pc->tile_x, pc->tile_y, pc->depth);
        }
}

// the kernel's form of the amplitude (spectrum_amplitude_fast: cheaper evaluation of the same formulas), n*n complex
void emul_spectrum_fast(int n, const SpectrumPC *pc, float *h0a) {
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) {
            const cplx a = spectrum_amplitude_fast(x, y, n, *pc);
            h0a[((size_t)y * n + x) * 2 + 0] = a.x;
            h0a[((size_t)y * n + x) * 2 + 1] = a.y;
        }
}

// one frame of one cascade: h0 = the stored half-spectrum plane (n*n complex); Tbuf = 4*n*n*2 floats (device layout, see t_unit); foam = n*n halves (device layout,
// Pass2::foam_index) is the recurrent state, read and rewritten; norm is written
int emul_frame(int n, const float *h0a, const float *omega, const CascadeFrame *cf, float *Tbuf, uint16_t *disp,
               uint16_t *norm, uint16_t *foam, float *f32) {
    switch (n) {
        case 128: frame<128>(h0a, omega, *cf, Tbuf, disp, norm, foam, f32); return 0;
        case 256: frame<256>(h0a, omega, *cf, Tbuf, disp, norm, foam, f32); return 0;
        case 512: frame<512>(h0a, omega, *cf, Tbuf, disp, norm, foam, f32); return 0;
        case 1024: frame<1024>(h0a, omega, *cf, Tbuf, disp, norm, foam, f32); return 0;
        case 2048: frame<2048>(h0a, omega, *cf, Tbuf, disp, norm, foam, f32); return 0;
    }
    return 1;
}

int emul_frame_lp(int n, int compact, const float *h0a, const float *omega, const CascadeFrame *cf, float *Tbuf, uint16_t *disp,
                  uint16_t *norm, uint16_t *foam, float *f32) {
    switch (n) {
        case 128: if (compact) return 1; frame<128>(h0a, omega, *cf, Tbuf, disp, norm, foam, f32, true); return 0;
        case 256: if (compact) frame_compact<256>(h0a, omega, *cf, Tbuf, disp, norm, foam, f32, true); else frame<256>(h0a, omega, *cf, Tbuf, disp, norm, foam, f32, true); return 0;
        case 512: if (compact) frame_compact<512>(h0a, omega, *cf, Tbuf, disp, norm, foam, f32, true); else frame<512>(h0a, omega, *cf, Tbuf, disp, norm, foam, f32, true); return 0;
    }
    return 1;
}

int emul_frame_compact(int n, const float *h0a, const float *omega, const CascadeFrame *cf, float *Tbuf, uint16_t *disp,
                       uint16_t *norm, uint16_t *foam, float *f32) {
    switch (n) {
        case 256: frame_compact<256>(h0a, omega, *cf, Tbuf, disp, norm, foam, f32); return 0;
        case 512: frame_compact<512>(h0a, omega, *cf, Tbuf, disp, norm, foam, f32); return 0;
        case 1024: frame_compact<1024>(h0a, omega, *cf, Tbuf, disp, norm, foam, f32); return 0;
    }
    return 1;
}

void emul_sincos(int count, const float *ph, float *sn, float *cs) {
    for (int i = 0; i < count; ++i) sincos_phase(ph[i], sn[i], cs[i]);
}
}
