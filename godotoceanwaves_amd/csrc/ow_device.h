// ow_device.h -- lane-level device code of the ocean-wave hot path (gfx950 / wave64).
//
// Everything here is written per LANE of one 64-wide wavefront; the kernels in
// ow_frame.hip / ow_spectrum.hip call these functions in sequence with wave-level
// synchronisation in between.  The same header also compiles as plain C++ (g++),
// which tests/emul/ uses to step 64 emulated lanes through the identical code on a
// machine without a GPU (test infrastructure; never part of the product path).
//
// Design: one wavefront owns one map row with all four packed
// spectra ("layers", spectrum_modulate.glsl:84-89).  A row transform of length N is a
// Stockham auto-sort DIF FFT, radix 16 x 16 x {-,2,4,8}, 16 points per lane held in
// registers, with two LDS exchanges between the radix passes; N/16 lanes cooperate on
// a row (so a wave carries 64/(N/16) rows when N < 1024).  No workgroup barrier exists
// anywhere: a workgroup IS one wave, and LDS operations of one wave execute in order.
#pragma once

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define OW_DEV __device__ __forceinline__
#define OW_HD __host__ __device__ __forceinline__
#define OW_DEVICE_BUILD 1
#else
#define OW_DEV inline
#define OW_HD inline
#define OW_DEVICE_BUILD 0
#endif

namespace ow {

struct alignas(8) cplx {
    float x, y;
};
struct alignas(16) f32x4 {
    float x, y, z, w;
};
struct alignas(8) u16x4 {
    uint16_t x, y, z, w;
};

constexpr float kPi = 3.141592653589793f;  // GLSL `#define PI` is an FP32 literal
constexpr float kG = 9.81f;                // GLSL `#define G`
constexpr int kLayers = 4;                 // NUM_SPECTRA (spectrum_modulate.glsl:14)

// ------------------------------------------------------------------------------------
// FFT plan (compile time).  N = 16 * T' ; T = lanes per row ; P = points per lane.
// ------------------------------------------------------------------------------------
constexpr int plan_T(int N) { return N >= 1024 ? 64 : N / 16; }
constexpr int plan_P(int N) { return N / plan_T(N); }
constexpr int plan_S(int N) { return N <= 256 ? 2 : 3; }
constexpr int plan_R(int N, int j) { return j == 0 ? 16 : (j == 1 ? (N == 128 ? 8 : 16) : N / 256); }
constexpr int plan_s(int N, int j) { return j == 0 ? 1 : plan_s(N, j - 1) * plan_R(N, j - 1); }  // stride
constexpr int plan_n(int N, int j) { return N / plan_s(N, j); }                                 // sub-length
constexpr int plan_m(int N, int j) { return plan_n(N, j) / plan_R(N, j); }
constexpr int plan_B(int N, int j) { return (N / plan_R(N, j)) / plan_T(N); }  // butterflies per lane
constexpr int plan_rows_per_wave(int N) { return 64 / plan_T(N); }
// pass 1 workgroup: as many waves as it takes to cover 4 consecutive rows (the row quad that forms one
// contiguous run of the transposed intermediate)
constexpr int plan_p1_waves(int N) { return plan_rows_per_wave(N) >= 4 ? 1 : 4 / plan_rows_per_wave(N); }
constexpr int plan_p1_rows(int N) { return plan_p1_waves(N) * plan_rows_per_wave(N); }
// LDS row region: FFT exchange image (N + N/16 complex) or staging image (N/2 float4), + 64 B so that the
// four row regions of a quad start on different 16-byte bank slots
constexpr int plan_region_cplx(int N) { return N + N / 16 + 8; }
// twiddle table: for every non-last stage j a [R_j - 1][m_j] block of exp(+2*pi*i*p*k/n_j)
constexpr int plan_tw_size(int N, int j) { return (plan_R(N, j) - 1) * plan_m(N, j); }
constexpr int plan_tw_off(int N, int j) { return j == 0 ? 0 : plan_tw_off(N, j - 1) + plan_tw_size(N, j - 1); }
constexpr int plan_tw_total(int N) { return plan_tw_off(N, plan_S(N) - 1); }
// LDS: one padded row buffer per row carried by the wave
constexpr int lds_slot(int e) { return e + (e >> 4); }
constexpr int plan_row_slots(int N) { return N + N / 16; }
constexpr int plan_lds_cplx(int N) { return plan_region_cplx(N) * plan_rows_per_wave(N); }

// register slot that holds output k of an in-place radix-R butterfly (see dft<R>)
constexpr int dft_pos(int R, int k) { return R == 16 ? 4 * (k % 4) + k / 4 : (R == 8 ? 2 * (k % 4) + k / 4 : k); }

// ------------------------------------------------------------------------------------
// complex helpers
// ------------------------------------------------------------------------------------
OW_DEV cplx cmul(cplx a, cplx b) { return cplx{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
OW_DEV cplx cadd(cplx a, cplx b) { return cplx{a.x + b.x, a.y + b.y}; }
OW_DEV cplx csub(cplx a, cplx b) { return cplx{a.x - b.x, a.y - b.y}; }
OW_DEV cplx cmuli(cplx a) { return cplx{-a.y, a.x}; }  // i * a

// Inverse-sign (e^{+2*pi*i/R}) radix butterflies, in place.
OW_DEV void dft2(cplx &a, cplx &b) {
    cplx t = csub(a, b);
    a = cadd(a, b);
    b = t;
}
OW_DEV void dft4(cplx &a, cplx &b, cplx &c, cplx &d) {
    cplx t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), t3 = cmuli(csub(b, d));
    a = cadd(t0, t2);
    b = cadd(t1, t3);
    c = csub(t0, t2);
    d = csub(t1, t3);
}

template <int R>
struct Dft;
template <>
struct Dft<2> {
    static OW_DEV void run(cplx *v) { dft2(v[0], v[1]); }
};
template <>
struct Dft<4> {
    static OW_DEV void run(cplx *v) { dft4(v[0], v[1], v[2], v[3]); }
};
template <>
struct Dft<8> {
    // n = 2*n1 + n2, k = k1 + 4*k2 : W8^{nk} = W4^{n1 k1} * W8^{n2 k1} * W2^{n2 k2}; output k at slot 2*k1 + k2
    static OW_DEV void run(cplx *v) {
        const float h = 0.70710678118654752f;
        dft4(v[0], v[2], v[4], v[6]);
        dft4(v[1], v[3], v[5], v[7]);
        v[3] = cplx{(v[3].x - v[3].y) * h, (v[3].x + v[3].y) * h};   // * W8^1
        v[5] = cmuli(v[5]);                                          // * W8^2
        v[7] = cplx{(-v[7].x - v[7].y) * h, (v[7].x - v[7].y) * h};  // * W8^3
        dft2(v[0], v[1]);
        dft2(v[2], v[3]);
        dft2(v[4], v[5]);
        dft2(v[6], v[7]);
    }
};
template <>
struct Dft<16> {
    // n = 4*n1 + n2, k = k1 + 4*k2 : W16^{nk} = W4^{n1 k1} * W16^{n2 k1} * W4^{n2 k2}; output k at slot 4*k1 + k2
    static OW_DEV void run(cplx *v) {
        const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
        dft4(v[0], v[4], v[8], v[12]);
        dft4(v[1], v[5], v[9], v[13]);
        dft4(v[2], v[6], v[10], v[14]);
        dft4(v[3], v[7], v[11], v[15]);
        // slot 4*k1 + n2 holds u[n2][k1]; multiply by W16^{n2*k1}
        v[5] = cmul(v[5], cplx{c1, s1});                                 // W16^1
        v[6] = cplx{(v[6].x - v[6].y) * h, (v[6].x + v[6].y) * h};       // W16^2
        v[7] = cmul(v[7], cplx{s1, c1});                                 // W16^3
        v[9] = cplx{(v[9].x - v[9].y) * h, (v[9].x + v[9].y) * h};       // W16^2
        v[10] = cmuli(v[10]);                                            // W16^4
        v[11] = cplx{(-v[11].x - v[11].y) * h, (v[11].x - v[11].y) * h}; // W16^6
        v[13] = cmul(v[13], cplx{s1, c1});                               // W16^3
        v[14] = cplx{(-v[14].x - v[14].y) * h, (v[14].x - v[14].y) * h}; // W16^6
        v[15] = cmul(v[15], cplx{-c1, -s1});                             // W16^9
        dft4(v[0], v[1], v[2], v[3]);
        dft4(v[4], v[5], v[6], v[7]);
        dft4(v[8], v[9], v[10], v[11]);
        dft4(v[12], v[13], v[14], v[15]);
    }
};

// ------------------------------------------------------------------------------------
// One row-FFT, lane view.  t = lane index inside the row (0..T-1), d = the lane's P points.
//   stage input layout : d[b*R + i]          <-> element q + s*(p + m*i),   u = t + T*b, q = u % s, p = u / s
//   stage output layout: d[b*R + dft_pos(k)] <-> element q + s*(R*p + k)    (times W_n^{p*k} unless last stage)
//   after the last stage: element index = t + T*b + (N/R)*k  (natural order, lanes contiguous)
// ------------------------------------------------------------------------------------
template <int N, int J>
OW_DEV void fft_stage_compute(cplx *d, int t, const cplx *__restrict__ tw) {
    constexpr int R = plan_R(N, J), B = plan_B(N, J), T = plan_T(N), s = plan_s(N, J), m = plan_m(N, J);
    constexpr bool last = (J == plan_S(N) - 1);
#pragma unroll
    for (int b = 0; b < B; ++b) {
        Dft<R>::run(d + b * R);
        if (!last) {
            const int p = (t + T * b) / s;
            const cplx *twj = tw + plan_tw_off(N, J);
#pragma unroll
            for (int k = 1; k < R; ++k) {
                d[b * R + dft_pos(R, k)] = cmul(d[b * R + dft_pos(R, k)], twj[(k - 1) * m + p]);
            }
        }
    }
}

template <int N, int J>
OW_DEV void fft_stage_write(const cplx *d, int t, cplx *lds_row) {
    constexpr int R = plan_R(N, J), B = plan_B(N, J), T = plan_T(N), s = plan_s(N, J);
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const int u = t + T * b, q = u % s, p = u / s;
#pragma unroll
        for (int k = 0; k < R; ++k) lds_row[lds_slot(q + s * (R * p + k))] = d[b * R + dft_pos(R, k)];
    }
}

template <int N, int J>
OW_DEV void fft_stage_read(cplx *d, int t, const cplx *lds_row) {
    constexpr int R = plan_R(N, J), B = plan_B(N, J), T = plan_T(N), s = plan_s(N, J), m = plan_m(N, J);
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const int u = t + T * b, q = u % s, p = u / s;
#pragma unroll
        for (int i = 0; i < R; ++i) d[b * R + i] = lds_row[lds_slot(q + s * (p + m * i))];
    }
}

// element index carried by register slot j before stage 0 / after the last stage
template <int N>
OW_DEV int fft_in_index(int t, int j) {
    constexpr int T = plan_T(N);
    return t + T * (j / 16) + (N / 16) * (j % 16);
}
template <int N>
struct OutMap {
    // register slot that holds natural output ordinal o (o-th element of this lane: index t + T*o ... see below)
    static constexpr int RL = plan_R(N, plan_S(N) - 1);
    static constexpr int BL = plan_B(N, plan_S(N) - 1);
    // slot (b, k) holds element t + T*b + (N/RL)*k ; N/RL = T*BL, so ordinal o = b + BL*k -> element t + T*o
    static constexpr int slot_of(int o) { return (o % BL) * RL + dft_pos(RL, o / BL); }
};

// ------------------------------------------------------------------------------------
// Accurate sin/cos of an FP32 phase up to ~1e5 rad (never the hardware approximations:
// SURVEY.md H1).  FP64 quadrant reduction (FP64 is half-rate on MI355X and the kernels are
// bandwidth bound), cephes minimax kernels on [-pi/4, pi/4]; max error ~1 ulp.
// ------------------------------------------------------------------------------------
OW_DEV void sincos_phase(float ph, float &sn, float &cs) {
    const double two_over_pi = 0.63661977236758134308, pio2 = 1.57079632679489661923;
    double pd = (double)ph;
    double q = __builtin_rint(pd * two_over_pi);
    float r = (float)__builtin_fma(-q, pio2, pd);
    int n = (int)q;
    float z = r * r;
    float s = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float c = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    float ss = (n & 1) ? c : s, cc = (n & 1) ? s : c;
    sn = (n & 2) ? -ss : ss;
    cs = ((n + 1) & 2) ? -cc : cc;
}

OW_DEV float fast_rcp(float x) {
#if OW_DEVICE_BUILD
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
OW_DEV float fast_sqrt(float x) {
#if OW_DEVICE_BUILD
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}
// IEEE FP32 product that the compiler may not fuse into a following add (phase = omega * t must be the
// FP32-rounded product, spectrum_modulate.glsl:65)
OW_DEV float mul_rn(float a, float b) {
#if OW_DEVICE_BUILD
    return __fmul_rn(a, b);
#else
    volatile float r = a * b;
    return r;
#endif
}

// float -> IEEE half bits, round to nearest even (RGBA16F image store)
OW_DEV uint16_t f2h(float f) {
#if OW_DEVICE_BUILD
    _Float16 h = (_Float16)f;  // v_cvt_f16_f32, RTE in the default mode
    return __builtin_bit_cast(uint16_t, h);
#else
    uint32_t x;
    __builtin_memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u, mag = x & 0x7FFFFFFFu;
    if (mag >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | ((mag > 0x7F800000u) ? 0x0200u : 0u));
    if (mag >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);
    if (mag < 0x38800000u) {
        if (mag < 0x33000000u) return (uint16_t)sign;
        int e = (int)(mag >> 23);
        uint32_t m = (mag & 0x7FFFFFu) | 0x800000u;
        int shift = 126 - e;
        uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (q & 1u))) q++;
        return (uint16_t)(sign | q);
    }
    uint32_t q = (((mag >> 23) - 112u) << 10) | ((mag & 0x7FFFFFu) >> 13), rem = mag & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (q & 1u))) q++;
    return (uint16_t)(sign | q);
#endif
}
OW_DEV float h2f(uint16_t h) {
#if OW_DEVICE_BUILD
    return (float)__builtin_bit_cast(_Float16, h);
#else
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu, x;
    if (e == 0) {
        float v = (float)m * 5.9604644775390625e-08f;
        __builtin_memcpy(&x, &v, 4);
        x |= sign;
    } else if (e == 31) {
        x = sign | 0x7F800000u | (m << 13);
    } else {
        x = sign | ((e + 112u) << 23) | (m << 13);
    }
    float f;
    __builtin_memcpy(&f, &x, 4);
    return f;
#endif
}

// ------------------------------------------------------------------------------------
// Per-cascade constants handed to the frame kernels (<= 8 cascades, MAX_CASCADES water.gdshader:8)
// ------------------------------------------------------------------------------------
struct CascadeFrame {
    float tile_x, tile_y;  // WaveCascadeParameters.tile_length
    float time;            // FP32-narrowed params.time (render_context.gd:131-134)
    float whitecap;
    float foam_grow_rate;
    float foam_decay;      // expf(-foam_decay_rate), evaluated once on the host (fft_unpack.glsl:62)
    int32_t cascade;       // which array layer / spectrum slot this launch slot works on
    int32_t pad0;
};
constexpr int kMaxCascades = 8;
struct FrameArgs {
    CascadeFrame c[kMaxCascades];
};

// Intermediate layout (device-private): two planes, one per packed layer PAIR p = layer/2,
//   T[c][p][y/8][x'][y%8] of float4 = (layer 2p, layer 2p+1) complex FP32   (16-byte units).
// One 128-byte line = 8 consecutive y of one x'.  Pass 1 stages a pair through LDS and every wave
// store instruction writes 16 x' x (4 rows x 16 B = one full 64-byte write request); pass 2 lane octets
// read whole 128-byte lines.
OW_HD size_t t_unit(int n, int pair, int xp, int y) {  // index in f32x4 units inside one cascade
    return ((((size_t)pair * (n >> 3) + (y >> 3)) * n + xp) << 3) + (y & 7);
}

// k-vector component exactly as spectrum_modulate.glsl:60 writes it
OW_DEV float modulate_kcomp(int id, int n, float tile) {
    return ((((float)id - (float)n * 0.5f) * 2.0f) * kPi) / tile;
}

// ------------------------------------------------------------------------------------
// PASS 1, lane view.  Row y of cascade c: load h0 + omega, time-modulate (spectrum_modulate.glsl:64-70),
// then for each packed layer build the row's spectrum and run the row IFFT (fft_compute.glsl, first
// dispatch); results go to the transposed intermediate T[c][x'][y][layer].
// ------------------------------------------------------------------------------------
template <int N>
struct Pass1 {
    static constexpr int T = plan_T(N), P = plan_P(N);

    // h[j] = h(k, t) for texel x = fft_in_index(t, j)
    static OW_DEV void load_modulate(cplx *h, int t, const f32x4 *__restrict__ h0_row,
                                     const float *__restrict__ om_row, float time) {
        f32x4 v[P];
        float om[P];
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int x = fft_in_index<N>(t, j);
            v[j] = h0_row[x];
            om[j] = om_row[x];
        }
#pragma unroll
        for (int j = 0; j < P; ++j) {
            float sn, cs;
            sincos_phase(mul_rn(om[j], time), sn, cs);
            // h = h0 * m + conj(h0(-k)) * conj(m),  m = (cs, sn)
            const float ar = v[j].x * cs - v[j].y * sn, ai = v[j].x * sn + v[j].y * cs;
            const float br = v[j].z * cs + v[j].w * sn, bi = v[j].w * cs - v[j].z * sn;
            h[j] = cplx{ar + br, ai + bi};
        }
    }

    // d[j] = packed layer L at texel x (spectrum_modulate.glsl:72-89); each layer is h times a complex
    // coefficient of the wave vector:  L0 = i(1+uy) h, L1 = (-ky + i ux) h, L2 = i(kx - ky uy) h,
    // L3 = -ux (kx + i ky) h.
    template <int L>
    static OW_DEV void layer_input(cplx *d, const cplx *h, int t, float ky, float dkx) {
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int x = fft_in_index<N>(t, j);
            const float kx = (float)(x - N / 2) * dkx;  // not phase-amplified: 1-2 ulp from :60 is harmless
            const float k = fast_sqrt(kx * kx + ky * ky) + 1e-6f;
            const float ik = fast_rcp(k);
            const float ux = kx * ik, uy = ky * ik;
            cplx cf = cplx{0.0f, 1.0f + uy};
            if (L == 1) cf = cplx{-ky, ux};
            if (L == 2) cf = cplx{0.0f, kx - ky * uy};
            if (L == 3) cf = cplx{-ux * kx, -ux * ky};
            d[j] = cmul(h[j], cf);
        }
    }

    // Transposed store of one layer pair (a = layer 2p, b = layer 2p+1), in two rounds r = 0, 1 that each
    // cover half of the x' range.  stage_write: every lane puts its own row's results, x'-ordered, into
    // its row region (16 B per x').  After a workgroup barrier, stage_store: thread tau of the block takes
    // (row q = tau % 4 of the quad, x' = tau / 4 + ...), so that consecutive lanes write consecutive
    // 16-byte units of T.  A second barrier must follow before the regions are written again.
    static OW_DEV void stage_write(const cplx *a, const cplx *b, int t, int r, cplx *lds_row) {
        f32x4 *st = reinterpret_cast<f32x4 *>(lds_row);
#pragma unroll
        for (int oo = 0; oo < P / 2; ++oo) {
            const int sl = OutMap<N>::slot_of(r * (P / 2) + oo);
            st[t + T * oo] = f32x4{a[sl].x, a[sl].y, b[sl].x, b[sl].y};
        }
    }
    // tau = thread index in the block, row0 = first map row of the block, Tc = this cascade's T
    static OW_DEV void stage_store(int tau, int r, int pair, int row0, const cplx *lds_block, f32x4 *__restrict__ Tc) {
        const int q = tau & 3, xi = (tau >> 2) % T, quad = tau / (4 * T);
        const int row = 4 * quad + q;
        const f32x4 *st = reinterpret_cast<const f32x4 *>(lds_block + row * plan_region_cplx(N));
#pragma unroll
        for (int k = 0; k < P / 2; ++k) {
            const int xl = xi + T * k;
            Tc[t_unit(N, pair, r * (N / 2) + xl, row0 + row)] = st[xl];
        }
    }
};

// ------------------------------------------------------------------------------------
// PASS 2, lane view.  Row x' of T: load, row IFFT of the 4 layers (fft_compute.glsl, second dispatch),
// then fft_unpack.glsl:38-68 fused: ifftshift sign, displacement, Jacobian/foam RMW, normal.
// ------------------------------------------------------------------------------------
template <int N>
struct Pass2 {
    static constexpr int T = plan_T(N), P = plan_P(N);

    // a = layer 2p, b = layer 2p+1 of row x' (16 bytes per y)
    static OW_DEV void load_pair(cplx *a, cplx *b, int t, int xp, int pair, const f32x4 *__restrict__ Tc) {
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const f32x4 v = Tc[t_unit(N, pair, xp, fft_in_index<N>(t, j))];
            a[j] = cplx{v.x, v.y};
            b[j] = cplx{v.z, v.w};
        }
    }

    // pair 0 (layers 0,1): displacement = (hx, hy, hz, 0) * sign (fft_unpack.glsl:44-50); dhy_dx is kept
    static OW_DEV void unpack_displacement(const cplx *l0, const cplx *l1, float *dhy_dx, int t, int xp,
                                           u16x4 *__restrict__ disp_row, float *__restrict__ f32_row) {
#pragma unroll
        for (int o = 0; o < P; ++o) {
            const int yp = t + T * o;
            const int sl = OutMap<N>::slot_of(o);
            const float sgn = ((xp ^ yp) & 1) ? -1.0f : 1.0f;  // fft_unpack.glsl:38
            const float hx = l0[sl].x * sgn, hy = l0[sl].y * sgn, hz = l1[sl].x * sgn;
            dhy_dx[o] = l1[sl].y * sgn;
            disp_row[yp] = u16x4{f2h(hx), f2h(hy), f2h(hz), f2h(0.0f * sgn)};
            if (f32_row) {
                float *q = f32_row + (size_t)yp * 8;
                q[0] = hx;
                q[1] = hy;
                q[2] = hz;
            }
        }
    }

    // pair 1 (layers 2,3): Jacobian, foam RMW, normalised slopes (fft_unpack.glsl:52-67)
    // f32_row (optional): 8 pre-quantisation channels per texel [hx,hy,hz,gx,gy,dhx_dx,foam,J]
    static OW_DEV void unpack_normal(const cplx *l2, const cplx *l3, const float *dhy_dx, int t, int xp,
                                     const CascadeFrame &cf, u16x4 *__restrict__ norm_row, float *__restrict__ f32_row) {
#pragma unroll
        for (int o = 0; o < P; ++o) {
            const int yp = t + T * o;
            const int sl = OutMap<N>::slot_of(o);
            const float sgn = ((xp ^ yp) & 1) ? -1.0f : 1.0f;
            const float dhy_dz = l2[sl].x * sgn, dhx_dx = l2[sl].y * sgn;
            const float dhz_dz = l3[sl].x * sgn, dhz_dx = l3[sl].y * sgn;

            const float jac = (1.0f + dhx_dx) * (1.0f + dhz_dz) - dhz_dx * dhz_dx;
            const float foam_factor = -fminf(0.0f, jac - cf.whitecap);
            float foam = h2f(norm_row[yp].w);
            foam = mul_rn(foam, cf.foam_decay);
            foam = foam + mul_rn(foam_factor, cf.foam_grow_rate);
            foam = fminf(fmaxf(foam, 0.0f), 1.0f);
            const float gx = dhy_dx[o] / (1.0f + fabsf(dhx_dx));
            const float gy = dhy_dz / (1.0f + fabsf(dhz_dz));

            norm_row[yp] = u16x4{f2h(gx), f2h(gy), f2h(dhx_dx), f2h(foam)};
            if (f32_row) {
                float *q = f32_row + (size_t)yp * 8;
                q[3] = gx;
                q[4] = gy;
                q[5] = dhx_dx;
                q[6] = foam;
                q[7] = jac;
            }
        }
    }
};

// Row IFFT of one layer held in d[] (lane view of the exchange points is in the callers: they must
// separate *_write and *_read with a wave-level sync).
template <int N>
struct RowFft {
    static constexpr int S = plan_S(N);
};

// ------------------------------------------------------------------------------------
// Spectrum initialisation, texel view (spectrum_compute.glsl, all of it) + the FP32 omega plane the
// frame kernels consume (spectrum_modulate.glsl:48-50,60-61 evaluated once: omega depends on
// tile_length and depth only).  Compiled with FP contraction OFF: omega must be bit-identical to the
// oracle's, which it is as long as +,*,/,sqrt are IEEE and tanh is correctly rounded.
// ------------------------------------------------------------------------------------
struct SpectrumPC {  // spectrum_compute.glsl:18-30
    int32_t seed_x, seed_y;
    float tile_x, tile_y;
    float alpha, peak_frequency, wind_speed, angle, depth, swell, detail, spread;
};

OW_DEV float omega_texel(int x, int y, int n, float tile_x, float tile_y, float depth) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const float kx = modulate_kcomp(x, n, tile_x), ky = modulate_kcomp(y, n, tile_y);
    const float k = sqrtf(kx * kx + ky * ky) + 1e-6f;
    const float a = k * depth;
    const float b = (float)tanh((double)a);  // correctly rounded tanhf
    return sqrtf(kG * k * b);
}

OW_DEV void hash_uniform(uint32_t x, uint32_t y, float &u1, float &u2) {  // spectrum_compute.glsl:34-41
    uint32_t h32 = y + 374761393u + x * 3266489917u;
    h32 = 2246822519u * (h32 ^ (h32 >> 15));
    h32 = 3266489917u * (h32 ^ (h32 >> 13));
    const uint32_t n = h32 ^ (h32 >> 16), n2 = n * 48271u;
    const float den = 2147483648.0f;  // float(0x7FFFFFFF)
    u1 = (float)((n >> 1) & 0x7FFFFFFFu) / den;
    u2 = (float)((n2 >> 1) & 0x7FFFFFFFu) / den;
}

OW_DEV cplx spectrum_amplitude(int idx, int idy, int n, const SpectrumPC &pc) {  // spectrum_compute.glsl:103-115
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const float dkx = (2.0f * kPi) / pc.tile_x, dky = (2.0f * kPi) / pc.tile_y;
    const float half = (float)n * 0.5f;
    const float kx = ((float)idx - half) * dkx, ky = ((float)idy - half) * dky;
    const float k = sqrtf(kx * kx + ky * ky) + 1e-6f;
    const float theta = atan2f(kx, ky);
    // dispersion_relation (:58-66)
    const float a = k * pc.depth, b = tanhf(a);
    const float w = sqrtf(kG * k * b);
    const float dw = (0.5f * kG) * (b + a * (1.0f - b * b)) / w;
    const float w_norm = dw / k * dkx * dky;
    // TMA_spectrum (:89-101)
    const float w_p = pc.peak_frequency;
    const float sigma = (w <= w_p) ? 0.07f : 0.09f;
    const float r = expf(-(w - w_p) * (w - w_p) / (2.0f * sigma * sigma * w_p * w_p));
    const float jonswap = (pc.alpha * kG * kG) / powf(w, 5.0f) * expf(-1.25f * powf(w_p / w, 4.0f)) * powf(3.3f, r);
    const float w_h = fminf(w * sqrtf(pc.depth / kG), 2.0f);
    const float kit = (w_h <= 1.0f) ? 0.5f * w_h * w_h : 1.0f - 0.5f * (2.0f - w_h) * (2.0f - w_h);
    const float s = jonswap * kit;
    // hasselmann_directional_spread (:81-86) + longuet_higgins (:69-78)
    const float pr = w / w_p;
    float sh = (w <= w_p) ? 6.97f * powf(fabsf(pr), 4.06f)
                          : 9.77f * powf(fabsf(pr), -2.33f - 1.45f * (pc.wind_speed * w_p / kG - 1.17f));
    sh = sh + 16.0f * tanhf(w_p / w) * pc.swell * pc.swell;
    const float sq = sqrtf(sh);
    const float lh_norm = (sh < 0.4f) ? (0.5f / kPi) + sh * (0.220636f + sh * (-0.109f + sh * 0.090f))
                                      : (1.0f / sqrtf(kPi)) * (sq * 0.5f + (1.0f / sq) * 0.0625f);
    const float hd = lh_norm * powf(fabsf(cosf((theta - pc.angle) * 0.5f)), 2.0f * sh);
    const float am = 1.0f - pc.spread;
    const float d = ((0.5f / kPi) * (1.0f - am) + hd * am) * expf(-(1.0f - pc.detail) * (1.0f - pc.detail) * k * k);
    // gaussian(hash(id + seed)) (:44-49)
    float u1, u2;
    hash_uniform((uint32_t)(idx + pc.seed_x), (uint32_t)(idy + pc.seed_y), u1, u2);
    const float rr = sqrtf(-2.0f * logf(u1)), th = (2.0f * kPi) * u2;
    const float amp = sqrtf(2.0f * s * d * w_norm);
    return cplx{rr * cosf(th) * amp, rr * sinf(th) * amp};
}

OW_DEV f32x4 spectrum_texel(int x, int y, int n, const SpectrumPC &pc) {  // spectrum_compute.glsl:117-125
    const cplx a = spectrum_amplitude(x, y, n, pc);
    const cplx b = spectrum_amplitude((n - x) % n, (n - y) % n, n, pc);
    return f32x4{a.x, a.y, b.x, -b.y};
}

}  // namespace ow
