// ow_device.h -- lane-level device code of the ocean-wave hot path (gfx950 / wave64).
//
// Everything here is written per LANE of one 64-wide wavefront; the kernels in
// ow_frame.hip / ow_spectrum.hip call these functions in sequence with wave-level
// synchronisation in between.  The same header also compiles as plain C++ (g++),
// which tests/emul/ uses to step 64 emulated lanes through the identical code on a
// machine without a GPU (test infrastructure; never part of the product path).
//
// Design: N/16 lanes own one map row, 16 points per lane, and transform its four packed
// spectra ("layers", spectrum_modulate.glsl:84-89) one after the other (or, in the layer-parallel kernels, one layer
// per lane group).  A row transform of length N is a Stockham auto-sort DIF FFT, radix 16 x 16 x {-,2,4,8}, butterflies
// in registers; the first exchange goes through LDS, the last one (N = 512, 1024) through the row-swap instructions.
// A wave carries 64/(N/16) rows when N < 1024; at N = 2048 a row spans two waves.  A workgroup is the lanes of 8
// consecutive rows; its barriers order LDS traffic only.
#pragma once

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define OW_DEV __device__ __forceinline__
#define OW_HD __host__ __device__ __forceinline__
#define OW_DEVICE_BUILD 1
// scheduling fence for the compiler only (no instruction): keeps live ranges short where it matters
#define OW_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define OW_DEV inline
#define OW_HD inline
#define OW_DEVICE_BUILD 0
#define OW_SCHED_FENCE() ((void)0)
#endif

namespace ow {

#if OW_DEVICE_BUILD
// two adjacent VGPRs: complex add / sub / scale map onto the packed FP32 instructions (v_pk_add_f32, v_pk_mul_f32,
// v_pk_fma_f32 with op_sel / neg modifiers) without register shuffling
typedef float cplx __attribute__((ext_vector_type(2)));
#else
struct alignas(8) cplx {
    float x, y;
};
#endif
struct alignas(16) f32x4 {
    float x, y, z, w;
};
struct alignas(8) u16x4 {
    uint16_t x, y, z, w;
};

// LDS read of one complex.  On the device the access is volatile in the LDS address space: the compiler then
// emits one ds_read_b64 per value (256 B/clk) instead of pairing them into ds_read2_b64 (128 B/clk).
OW_DEV cplx lds_read(const cplx *p) {
#if OW_DEVICE_BUILD
    typedef __attribute__((address_space(3))) cplx lds_cplx;
    return *(const volatile lds_cplx *)p;
#else
    return *p;
#endif
}

constexpr float kPi = 3.141592653589793f;  // GLSL `#define PI` is an FP32 literal
constexpr float kG = 9.81f;                // GLSL `#define G`
constexpr int kLayers = 4;                 // NUM_SPECTRA (spectrum_modulate.glsl:14)

// ------------------------------------------------------------------------------------
// FFT plan (compile time).  P = 16 points per lane for every size; T = N/16 lanes cooperate on a row
// (T = 128 at N = 2048: the row's exchanges then cross two waves and use the workgroup barrier).
// ------------------------------------------------------------------------------------
constexpr int kP = 16;
constexpr int plan_T(int N) { return N / kP; }
constexpr int plan_P(int) { return kP; }
constexpr int plan_S(int N) { return N <= 256 ? 2 : 3; }
constexpr int plan_R(int N, int j) { return j == 0 ? 16 : (j == 1 ? (N == 128 ? 8 : 16) : N / 256); }
constexpr int plan_s(int N, int j) { return j == 0 ? 1 : (j == 1 ? 16 : 16 * plan_R(N, 1)); }  // stride = product of earlier radices (no recursion: must fold)
constexpr int plan_n(int N, int j) { return N / plan_s(N, j); }                                 // sub-length
constexpr int plan_m(int N, int j) { return plan_n(N, j) / plan_R(N, j); }
constexpr int plan_B(int N, int j) { return (N / plan_R(N, j)) / plan_T(N); }  // butterflies per lane
// workgroup = the lanes of 8 consecutive rows.  Pass 1: each x' gets one full 64-byte write request of the
// transposed intermediate; both passes: the 8 rows share one LDS copy of the twiddle table.
constexpr int kWgRows = 8;
constexpr int plan_wg_threads(int N) { return kWgRows * plan_T(N); }
constexpr bool plan_row_spans_waves(int N) { return plan_T(N) > 64; }
// LDS row region: FFT exchange image (N + N/16 complex, padded slots) or staging image (N complex, linear),
// + 32 B so that the 8 row regions of a block start 8 banks apart (conflict-free transposed reads)
constexpr int plan_region_cplx(int N) { return N + N / 16 + 4; }
// twiddle table: for every non-last stage j a [R_j - 1][m_j] block of exp(+2*pi*i*p*k/n_j)
constexpr int plan_tw_size(int N, int j) { return (plan_R(N, j) - 1) * plan_m(N, j); }
constexpr int plan_tw_off(int N, int j) { return j == 0 ? 0 : (j == 1 ? plan_tw_size(N, 0) : plan_tw_size(N, 0) + plan_tw_size(N, 1)); }
constexpr int plan_tw_total(int N) { return plan_tw_off(N, plan_S(N) - 1); }
// LDS slot of element e of a row's exchange image: e + (e >> 4) -- one padding slot per 16 elements.
// The LDS rules that matter (MI355X_MICROARCH.md, LDS): a wave's ds_write_b64 is served in four groups of 16 CONTIGUOUS lanes over 32 banks
// -- conflict-free when the 16 slots (8-byte units) of a group are distinct mod 16 --, a ds_read_b64 in two groups of 32 lanes over 64 banks
// -- distinct mod 32.  Under this map the stage-0 writes (element 16 t + k over lanes t: 17 t + k) are conflict-free and every stage-1 read
// (element q + 16 p + 64 i over lanes (q, p): the runs q + 17 p of p = 0 and p = 1 share one slot mod 32) takes two passes per group instead
// of one: +32 passes per 1024-point transform, the 14 % of LDS-active cycles that round 2's SQ counters report as bank conflicts.
// Round 3 went after them (tools/lds_bank_model.py, profiles/r03_pmc_lds_counters.txt, profiles/r03_lds_padding_ab.txt):
//   map 1, e + (e >> 5): frees the reads, two-way conflicts on every stage-0 write -- SQ_LDS_BANK_CONFLICT of k_tick_pair_c<1024> 17.05 M ->
//     34.09 M (it came out of a first model that gave writes the reads' lane groups; the counter corrected the model);
//   map 2, bits 3 and 4 of e swapped, + (e >> 5) (at 2048, map 1 for the second exchange, whose region is rewritten in between): the model's
//     answer with the right groups -- SQ_LDS_BANK_CONFLICT = 0 for k_tick_pair_c<1024> AND for k_pass2c<2048> (24.1 M before), LDS-active cycles
//     -14 % / -19 %.  Results bit-identical (the map is private to a row's exchange).
// Time did not move for any of them, at any size, on the same box (1024^2 x 4 53.5 vs 53.6 us, 2048^2 x 1 63.3 vs 63.1): the LDS is not on
// the critical path of these kernels.  And map 2's lane part costs eight integer operations per stage where this one costs three, which
// pushes k_pass1c_split<2048> and k_tick_group_c_lp<256> from 126 to 128 VGPRs plus 16 - 20 bytes of scratch.  A change that buys nothing
// and spills is not shipped: the map of rounds 1-2 stays, the others remain selectable for A/B builds (profiles/EXPERIMENTS.md).
#ifndef OW_LDS_SLOT_MAP
#define OW_LDS_SLOT_MAP 0  // 0: e + (e >> 4) (shipped), 1: e + (e >> 5), 2: bits 3 / 4 swapped + (e >> 5), second exchange map 1   (scripts/build_variant.sh mapN -DOW_LDS_SLOT_MAP=N)
#endif
// X = which exchange of the transform (0: between stages 0 and 1; 1: between stages 1 and 2, which goes through LDS at 2048 only)
template <int X>
constexpr int lds_slot(int e) {
    return OW_LDS_SLOT_MAP == 0 ? e + (e >> 4)
           : (OW_LDS_SLOT_MAP == 1 || (OW_LDS_SLOT_MAP == 2 && X == 1)) ? e + (e >> 5)
                                                                        : ((e & ~0x18) | (((e >> 3) & 1) << 4) | (((e >> 4) & 1) << 3)) + (e >> 5);
}
// whole workgroup: [twiddle table][8 x row region]
constexpr int plan_wg_lds_cplx(int N) { return plan_region_cplx(N) * kWgRows + plan_tw_total(N); }

// register slot that holds output k of an in-place radix-R butterfly (see dft<R>)
constexpr int dft_pos(int R, int k) { return R == 16 ? 4 * (k % 4) + k / 4 : (R == 8 ? 2 * (k % 4) + k / 4 : k); }

// ------------------------------------------------------------------------------------
// complex helpers
// ------------------------------------------------------------------------------------
#if OW_DEVICE_BUILD
// Every helper below is ONE or TWO packed instructions: the half swap of "times i" rides on op_sel and the
// lane-wise sign on a (+-1, -+1) constant pair, instead of v_xor + v_mov in front of a packed add.
OW_DEV cplx cadd(cplx a, cplx b) { return a + b; }
OW_DEV cplx csub(cplx a, cplx b) { return a - b; }
OW_DEV cplx cmuli(cplx a) { return a.yx * cplx{-1.0f, 1.0f}; }                                    // i * a
OW_DEV cplx caddi(cplx a, cplx b) { return __builtin_elementwise_fma(b.yx, cplx{-1.0f, 1.0f}, a); }  // a + i * b
OW_DEV cplx csubi(cplx a, cplx b) { return __builtin_elementwise_fma(b.yx, cplx{1.0f, -1.0f}, a); }  // a - i * b
OW_DEV cplx cscale(cplx a, float s) { return a * s; }
// a * (c + i s) with compile-time c, s
OW_DEV cplx cmul_const(cplx a, float c, float sn) { return __builtin_elementwise_fma(a.yx, cplx{-sn, sn}, a * cplx{c, c}); }
// a * b, general: (a.x b.x, a.x b.y) + a.y * (-b.y, b.x).  The sign of the one half rides on the packed instruction's neg_lo modifier, which the
// compiler does not use for a per-half sign (it emits a third instruction, a.yy * (-1, 1)); spelled out, a complex multiply is TWO packed
// instructions -- thirty fewer per 1024-point transform (round 5).  Same products, same fma: bit-identical.
// (the two inline-asm spellings of this header -- this one and v_cvt_pk_f16_f32 in f2h2 -- are gfx950 encodings: any other --offload-arch takes the
//  C forms below them instead of failing in the assembler; the library itself is built for gfx950 only, build.py)
#if !defined(OW_CMUL_NEG_LO) && defined(__gfx950__)
#define OW_CMUL_NEG_LO 1
#elif !defined(OW_CMUL_NEG_LO)
#define OW_CMUL_NEG_LO 0
#endif
OW_DEV cplx cmul(cplx a, cplx b) {
#if OW_CMUL_NEG_LO
    const cplx t = a.xx * b;
    cplx r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
#else
    return __builtin_elementwise_fma(a.yy * cplx{-1.0f, 1.0f}, b.yx, a.xx * b);
#endif
}
#else
OW_DEV cplx cadd(cplx a, cplx b) { return cplx{a.x + b.x, a.y + b.y}; }
OW_DEV cplx csub(cplx a, cplx b) { return cplx{a.x - b.x, a.y - b.y}; }
OW_DEV cplx cmuli(cplx a) { return cplx{-a.y, a.x}; }  // i * a
OW_DEV cplx caddi(cplx a, cplx b) { return cplx{a.x - b.y, a.y + b.x}; }
OW_DEV cplx csubi(cplx a, cplx b) { return cplx{a.x + b.y, a.y - b.x}; }
OW_DEV cplx cscale(cplx a, float s) { return cplx{a.x * s, a.y * s}; }
OW_DEV cplx cmul_const(cplx a, float c, float sn) { return cplx{a.x * c - a.y * sn, a.y * c + a.x * sn}; }
OW_DEV cplx cmul(cplx a, cplx b) { return cplx{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
#endif

// Inverse-sign (e^{+2*pi*i/R}) radix butterflies, in place.
OW_DEV void dft2(cplx &a, cplx &b) {
    cplx t = csub(a, b);
    a = cadd(a, b);
    b = t;
}
OW_DEV void dft4(cplx &a, cplx &b, cplx &c, cplx &d) {
    const cplx t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), t3 = csub(b, d);
    a = cadd(t0, t2);
    b = caddi(t1, t3);
    c = csub(t0, t2);
    d = csubi(t1, t3);
}
constexpr float kH = 0.70710678118654752f;
OW_DEV cplx cmul_w8_1(cplx a) { return cmul_const(a, kH, kH); }   // a * (1 + i) / sqrt(2)
OW_DEV cplx cmul_w8_3(cplx a) { return cmul_const(a, -kH, kH); }  // a * (-1 + i) / sqrt(2)

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
        dft4(v[0], v[2], v[4], v[6]);
        dft4(v[1], v[3], v[5], v[7]);
        v[3] = cmul_w8_1(v[3]);  // * W8^1
        v[5] = cmuli(v[5]);      // * W8^2
        v[7] = cmul_w8_3(v[7]);  // * W8^3
        dft2(v[0], v[1]);
        dft2(v[2], v[3]);
        dft2(v[4], v[5]);
        dft2(v[6], v[7]);
    }
};
template <>
struct Dft<16> {
    // n = 4*n1 + n2, k = k1 + 4*k2 : W16^{nk} = W4^{n1 k1} * W16^{n2 k1} * W4^{n2 k2}; output k at slot 4*k1 + k2
    // (scheduling fences keep at most two radix-4 butterflies' temporaries alive at a time)
    static OW_DEV void run(cplx *v) {
        const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f;
        dft4(v[0], v[4], v[8], v[12]);
        dft4(v[1], v[5], v[9], v[13]);
        OW_SCHED_FENCE();
        dft4(v[2], v[6], v[10], v[14]);
        dft4(v[3], v[7], v[11], v[15]);
        OW_SCHED_FENCE();
        // slot 4*k1 + n2 holds u[n2][k1]; multiply by W16^{n2*k1}
        v[5] = cmul_const(v[5], c1, s1);           // W16^1
        v[6] = cmul_w8_1(v[6]);                    // W16^2
        v[7] = cmul_const(v[7], s1, c1);           // W16^3
        v[9] = cmul_w8_1(v[9]);                    // W16^2
        v[10] = cmuli(v[10]);                      // W16^4
        v[11] = cmul_w8_3(v[11]);                  // W16^6
        v[13] = cmul_const(v[13], s1, c1);         // W16^3
        v[14] = cmul_w8_3(v[14]);                  // W16^6
        v[15] = cmul_const(v[15], -c1, -s1);       // W16^9
        OW_SCHED_FENCE();
        dft4(v[0], v[1], v[2], v[3]);
        dft4(v[4], v[5], v[6], v[7]);
        OW_SCHED_FENCE();
        dft4(v[8], v[9], v[10], v[11]);
        dft4(v[12], v[13], v[14], v[15]);
        OW_SCHED_FENCE();
    }
};

// ------------------------------------------------------------------------------------
// One row-FFT, lane view.  t = lane index inside the row (0..T-1), d = the lane's P points.
//   stage input layout : d[b*R + i]          <-> element q + s*(p + m*i),   u = t + T*b, q = u % s, p = u / s
//   stage output layout: d[b*R + dft_pos(k)] <-> element q + s*(R*p + k)    (times W_n^{p*k} unless last stage)
//   after the last stage: element index = t + T*b + (N/R)*k  (natural order, lanes contiguous)
// ------------------------------------------------------------------------------------
// exp(+2 pi i m / 32) for compile-time m in [0, 16] (the ordinal part of W_N^k; the lane part comes from a table)
constexpr float kRootCos32[9] = {1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f,
                                 0.55557023301960218f, 0.38268343236508977f, 0.19509032201612825f, 0.0f};
constexpr float root32_cos(int i) { return i <= 8 ? kRootCos32[i] : -kRootCos32[16 - i]; }
constexpr float root32_sin(int i) { return i <= 8 ? kRootCos32[8 - i] : kRootCos32[i - 8]; }
// HALF TABLE (TWH, rows that span two waves: N = 2048).  The stage-0 block of the table is [15][N/16] = 15 K complex at N = 2048, and with
// it an 8-wave block of pass 2 (4 columns) needs 86 KB of LDS: two of them do not fit a CU.  But lane t >= 64 -- the row's second wave --
// needs exp(2 pi i t k / N) = exp(2 pi i (t - 64) k / N) * exp(2 pi i k / 32): the first wave's entry times a compile-time 32nd root of
// unity.  So the table keeps the first 64 columns only ([15][64], then the stage-1 block unchanged: plan_twh_total), 8.6 KB, and the second
// wave of a row pays fifteen constant rotations per transform (wave-uniform branch).  78 KB per block: two blocks per CU, of either pass.
constexpr int kTwhCols = 64;
constexpr int plan_twh_off(int N, int j) { return j == 0 ? 0 : (plan_R(N, 0) - 1) * kTwhCols; }
constexpr int plan_twh_total(int N) { return plan_twh_off(N, 1) + plan_tw_size(N, 1); }
template <int N, int J, bool TWH = false>
OW_DEV void fft_stage_compute(cplx *d, int t, const cplx *__restrict__ tw) {
    constexpr int R = plan_R(N, J), B = plan_B(N, J), T = plan_T(N), s = plan_s(N, J), m = plan_m(N, J);
    constexpr bool last = (J == plan_S(N) - 1);
    static_assert(!TWH || (plan_S(N) == 3 && plan_B(N, 0) == 1 && plan_T(N) == 2 * kTwhCols), "half table: a row of exactly two waves");
#pragma unroll
    for (int b = 0; b < B; ++b) Dft<R>::run(d + b * R);
    if (!last) {
        // twiddles are fetched (LDS table) only now, when the butterfly's temporaries are dead
        OW_SCHED_FENCE();
        constexpr int off = TWH ? plan_twh_off(N, J) : plan_tw_off(N, J);
        const cplx *twj = tw + off;
        if constexpr (TWH && J == 0) {
#if OW_DEVICE_BUILD
            const bool upper = __builtin_amdgcn_readfirstlane(t) >= kTwhCols;  // the row's second wave (wave-uniform)
#else
            const bool upper = t >= kTwhCols;
#endif
            const int p = t & (kTwhCols - 1);
            if (upper) {
#pragma unroll
                for (int k = 1; k < R; ++k) d[dft_pos(R, k)] = cmul_const(d[dft_pos(R, k)], root32_cos(k), root32_sin(k));
            }
#pragma unroll
            for (int k = 1; k < R; ++k) {
                d[dft_pos(R, k)] = cmul(d[dft_pos(R, k)], lds_read(twj + (k - 1) * kTwhCols + p));
                if (k % 5 == 0) OW_SCHED_FENCE();
            }
        } else {
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const int p = (t + T * b) / s;
#pragma unroll
                for (int k = 1; k < R; ++k) {
                    d[b * R + dft_pos(R, k)] = cmul(d[b * R + dft_pos(R, k)], lds_read(twj + (k - 1) * m + p));
                    if (k % 5 == 0) OW_SCHED_FENCE();  // a few table reads in flight at a time, not all 15
                }
            }
        }
        OW_SCHED_FENCE();
    }
}

// LDS slot of the element a lane writes after / reads before a stage.  For every plan used here the slot is
// affine in (b, k): slot(t, b, k) = slot(t, 0, 0) + [slot(0, b, k) - slot(0, 0, 0)], because T is a multiple of
// 16 (or, at N = 128, the lane index stays below 16): the lane part and the (b, k) part of the element index occupy disjoint
// bits, and every map of lds_slot (a bit permutation plus a shifted copy) is additive over numbers with disjoint bits.  The lane part is computed once per stage; the (b, k) part is a compile-time DS offset.
template <int N, int J>
constexpr int wr_slot(int t, int b, int k) {
    const int R = plan_R(N, J), T = plan_T(N), s = plan_s(N, J);
    const int u = t + T * b, q = u % s, p = u / s;
    return lds_slot<J>(q + s * (R * p + k));
}
template <int N, int J>
constexpr int rd_slot(int t, int b, int i) {
    const int T = plan_T(N), s = plan_s(N, J), m = plan_m(N, J);
    const int u = t + T * b, q = u % s, p = u / s;
    return lds_slot<J - 1>(q + s * (p + m * i));
}

template <int N, int J>
OW_DEV void fft_stage_write(const cplx *d, int t, cplx *lds_row) {
    constexpr int R = plan_R(N, J), B = plan_B(N, J);
    cplx *base = lds_row + wr_slot<N, J>(t, 0, 0);
#pragma unroll
    for (int b = 0; b < B; ++b) {
#pragma unroll
        for (int k = 0; k < R; ++k) base[wr_slot<N, J>(0, b, k) - wr_slot<N, J>(0, 0, 0)] = d[b * R + dft_pos(R, k)];
    }
}

template <int N, int J>
OW_DEV void fft_stage_read(cplx *d, int t, const cplx *lds_row) {
    constexpr int R = plan_R(N, J), B = plan_B(N, J);
    const cplx *base = lds_row + rd_slot<N, J>(t, 0, 0);
#pragma unroll
    for (int b = 0; b < B; ++b) {
#pragma unroll
        for (int i = 0; i < R; ++i) d[b * R + i] = lds_read(base + (rd_slot<N, J>(0, b, i) - rd_slot<N, J>(0, 0, 0)));
    }
}

// Exchange between stage 1 and the last stage WITHOUT LDS, for N = 512 and 1024 (a row inside one wave).
// After stage 1, lane (q = t%16, p = t/16) holds output k of its radix-16 butterfly = element q + 256*p + 16*k;
// the last stage (radix R2 = N/256) wants, in lane (q, r) and butterfly b, the R2 elements q + 16*r + 64*b... i.e.
// with k = R2*b + r:  (lane row p, register k)  ->  (lane row r, input p of butterfly b).  For every b that is an
// R2 x R2 transpose between the 16-lane row index and the register index, done with the gfx950 row-swap
// instructions (v_permlane32_swap: upper 32 lanes of A <-> lower 32 lanes of B; v_permlane16_swap: odd 16-lane
// rows of A <-> even rows of B) on the VALU of the wave's own SIMD instead of through the CU-wide LDS pipe.
constexpr bool plan_lane_exchange(int N) { return N == 512 || N == 1024; }
#if OW_DEVICE_BUILD
OW_DEV void swap_rows32(cplx &a, cplx &b) {
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    // (element values are copied to scalars first: bit-casting the vector-element lvalue directly miscompiles)
    const float ax = a.x, ay = a.y, bx = b.x, by = b.y;
    const u2 rx = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, ax), __builtin_bit_cast(uint32_t, bx), false, false);
    const u2 ry = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, ay), __builtin_bit_cast(uint32_t, by), false, false);
    const uint32_t nax = rx.x, nbx = rx.y, nay = ry.x, nby = ry.y;
    a = cplx{__builtin_bit_cast(float, nax), __builtin_bit_cast(float, nay)};
    b = cplx{__builtin_bit_cast(float, nbx), __builtin_bit_cast(float, nby)};
}
OW_DEV void swap_rows16(cplx &a, cplx &b) {
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    // (element values are copied to scalars first: bit-casting the vector-element lvalue directly miscompiles)
    const float ax = a.x, ay = a.y, bx = b.x, by = b.y;
    const u2 rx = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, ax), __builtin_bit_cast(uint32_t, bx), false, false);
    const u2 ry = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, ay), __builtin_bit_cast(uint32_t, by), false, false);
    const uint32_t nax = rx.x, nbx = rx.y, nay = ry.x, nby = ry.y;
    a = cplx{__builtin_bit_cast(float, nax), __builtin_bit_cast(float, nay)};
    b = cplx{__builtin_bit_cast(float, nbx), __builtin_bit_cast(float, nby)};
}
#endif
// Order of the swaps and the final renaming, shared by the device code and the 64-lane CPU emulation: SW32 / SW16
// are called as SW(slot_a, slot_b) for whole-wave registers.
template <int N, class SW32, class SW16, class MOVE>
OW_HD void lane_exchange_plan(SW32 sw32, SW16 sw16, MOVE move) {
    constexpr int R2 = plan_R(N, 2), NB = 16 / R2;
    for (int b = 0; b < NB; ++b) {
        int X[4];
        for (int r = 0; r < R2; ++r) X[r] = dft_pos(16, R2 * b + r);
        if (R2 == 4) {
            sw32(X[0], X[2]);
            sw32(X[1], X[3]);
            sw16(X[0], X[1]);
            sw16(X[2], X[3]);
        } else {
            sw16(X[0], X[1]);
        }
        for (int p = 0; p < R2; ++p) move(b * R2 + p, X[p]);  // last-stage input p of butterfly b <- register X_p
    }
}
#if OW_DEVICE_BUILD
template <int N>
OW_DEV void fft_lane_exchange(cplx *d) {
    constexpr int R2 = plan_R(N, 2), NB = 16 / R2;
    cplx o[kP];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        cplx *X[4];
#pragma unroll
        for (int r = 0; r < R2; ++r) X[r] = d + dft_pos(16, R2 * b + r);
        if constexpr (R2 == 4) {
            swap_rows32(*X[0], *X[2]);
            swap_rows32(*X[1], *X[3]);
            swap_rows16(*X[0], *X[1]);
            swap_rows16(*X[2], *X[3]);
        } else {
            swap_rows16(*X[0], *X[1]);
        }
#pragma unroll
        for (int p = 0; p < R2; ++p) o[b * R2 + p] = *X[p];
    }
#pragma unroll
    for (int j = 0; j < kP; ++j) d[j] = o[j];
}
#endif

// element index carried by register slot j before stage 0: t + T*j  (N/16 == T)
template <int N>
constexpr int fft_in_step(int j) {
    return plan_T(N) * j;
}
template <int N>
OW_DEV int fft_in_index(int t, int j) {
    return t + fft_in_step<N>(j);
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
// SPLIT PLAN for pass 1 of rows that do not fit one wave at 16 points per lane (N = 2048): decimation in time,
//   X[k] = E[k] + W_N^k O[k],   X[k + N/2] = E[k] - W_N^k O[k],   E / O = transforms of the even / odd elements (length N/2),
// one WAVE per parity.  Each wave runs the N/2 plan (one LDS exchange inside the wave, the last exchange on the row-swap
// instructions): no rendezvous with the partner wave during the transform.  The final radix-2 step needs both waves' results,
// and pass 1 stages its results in LDS for the transposed store anyway: both waves stage their half and the STORING threads
// combine E and O on their way out, so the two waves of a row never have to meet.
// Physical lane (parity w, lane tp) plays the LOGICAL lane t = 2 tp + w of the N plan: its texels are t + (N/16) rot(j), i.e.
// all input-side lane code (loads, modulation, wave numbers, layer inputs, the Nyquist-line special cases) is the N plan's,
// called with the logical lane.  The (-1)^x' half of the ifftshift is a shift by N/2 of the full sequence = N/4 of each parity
// subsequence = 8 lane-strides of the N/2 plan: the same slot rotation.
// (Pass 2 keeps the N plan: measured, the same split there -- half exchange between partner lanes, two rendezvous per transform
// instead of four -- is no faster: 33.7 against 32.3 us at 2048^2 x 1, 34.9 against 35.2 at x 4.)
// ------------------------------------------------------------------------------------
constexpr bool plan_split(int N) { return N == 2048; }

// ------------------------------------------------------------------------------------
// Accurate sin/cos of an FP32 phase up to ~2.5e4 rad (never the hardware approximations: SURVEY.md H1).
// Three-step Cody-Waite reduction by pi in FP32 with FMA (pi = P1 + P2 + P3, P1 8 bits and P2 11 bits so that
// n*P1 and n*P2 are exact for n < 2^13 and the first two subtractions cancel exactly), then minimax polynomials
// on [-pi/2, pi/2] (sin: odd, degree 9; cos: even, degree 10) and the sign (-1)^n on both.  Measured error
// <= 1.3e-7 (sin), 0.8e-7 (cos) absolute: 1 ulp at 1.0.  No FP64, no table, no branch.
// ------------------------------------------------------------------------------------
OW_DEV void sincos_phase(float ph, float &sn, float &cs) {
    const float n = __builtin_rintf(ph * 0.318309886183790672f);
    float r = __builtin_fmaf(-n, 3.140625f, ph);
    r = __builtin_fmaf(-n, 9.67502593994140625e-4f, r);
    r = __builtin_fmaf(-n, 1.509957990978376432e-7f, r);
    const float z = r * r;
    float p = 2.6343420813645935e-06f;
    p = __builtin_fmaf(p, z, -0.00019822614558506757f);
    p = __builtin_fmaf(p, z, 0.008333241567015648f);
    p = __builtin_fmaf(p, z, -0.1666666567325592f);
    const float s = __builtin_fmaf(p * z, r, r);
    float q = -2.654252000411361e-07f;
    q = __builtin_fmaf(q, z, 2.478597525623627e-05f);
    q = __builtin_fmaf(q, z, -0.0013888811226934195f);
    q = __builtin_fmaf(q, z, 0.0416666679084301f);
    const float c = __builtin_fmaf(q * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
    const uint32_t flip = (uint32_t)(int)n << 31;  // (-1)^n as a sign bit
    sn = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, s) ^ flip);
    cs = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, c) ^ flip);
}
// The same, returned as m = (cos, sin) = exp(i ph) for the packed complex arithmetic.  On the device the two polynomials -- independent Horner
// chains in the same z -- run as ONE chain of packed instructions (v_pk_fma_f32: each half is the IEEE fma of the scalar form, so the
// results are bit-identical to sincos_phase; round 5: eleven scalar fma / mul per texel become five packed ones).
#ifndef OW_SINCOS_PACKED
#define OW_SINCOS_PACKED 1
#endif
OW_DEV cplx expi_phase(float ph) {
#if OW_DEVICE_BUILD && OW_SINCOS_PACKED
    const float n = __builtin_rintf(ph * 0.318309886183790672f);
    float r = __builtin_fmaf(-n, 3.140625f, ph);
    r = __builtin_fmaf(-n, 9.67502593994140625e-4f, r);
    r = __builtin_fmaf(-n, 1.509957990978376432e-7f, r);
    const float z = r * r;
    const cplx zz = cplx{z, z};
    cplx p = cplx{-2.654252000411361e-07f, 2.6343420813645935e-06f};  // (cos chain, sin chain)
    p = __builtin_elementwise_fma(p, zz, cplx{2.478597525623627e-05f, -0.00019822614558506757f});
    p = __builtin_elementwise_fma(p, zz, cplx{-0.0013888811226934195f, 0.008333241567015648f});
    p = __builtin_elementwise_fma(p, zz, cplx{0.0416666679084301f, -0.1666666567325592f});
    p = p * zz;
    // cos = (q z) z + (1 - z / 2),  sin = (p z) r + r
    const cplx m = __builtin_elementwise_fma(p, cplx{z, r}, cplx{__builtin_fmaf(-0.5f, z, 1.0f), r});
    const uint32_t flip = (uint32_t)(int)n << 31;  // (-1)^n as a sign bit
    const float mx = m.x, my = m.y;
    return cplx{__builtin_bit_cast(float, __builtin_bit_cast(uint32_t, mx) ^ flip), __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, my) ^ flip)};
#else
    float sn, cs;
    sincos_phase(ph, sn, cs);
    return cplx{cs, sn};
#endif
}

// returns x, but the compiler cannot see that: stops it from keeping (instead of recomputing) cheap
// per-texel terms across long code regions
OW_DEV float opaque(float x) {
#if OW_DEVICE_BUILD
    asm volatile("" : "+v"(x));
#endif
    return x;
}
// redefines c "in place" as far as the compiler can tell: nothing derived from the old value (swizzled or
// negated copies for the packed instructions) is carried past this point
OW_DEV void opaque_inplace(cplx &c) {
#if OW_DEVICE_BUILD
    asm volatile("" : "+v"(c));
#endif
}
OW_DEV int opaque(int x) {
#if OW_DEVICE_BUILD
    asm volatile("" : "+v"(x));
#endif
    return x;
}
OW_DEV float fast_rcp(float x) {
#if OW_DEVICE_BUILD
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
OW_DEV float fast_rsq(float x) {
#if OW_DEVICE_BUILD
    return __builtin_amdgcn_rsqf(x);
#else
    return 1.0f / sqrtf(x);
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
    // The conversion is spelled as the instruction (RTE in the default mode): written as a C cast, the compiler fuses
    // a producing multiply into it (v_fma_mixlo_f16: ONE rounding of the exact product), whereas the quantity the
    // reference stores is the FP32-rounded value (imageStore of an FP32 result) -- a 1-ulp(FP16) difference in rare cases.
    uint32_t r;
    asm("v_cvt_f16_f32_e32 %0, %1" : "=v"(r) : "v"(f));
    return (uint16_t)r;
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
// two floats -> two IEEE halves in one word (lo | hi << 16), each rounded to nearest even exactly like f2h.  gfx950 has the packed conversion
// (v_cvt_pk_f16_f32: same rounding mode, same denormal handling as v_cvt_f16_f32 -- checked over all 2^32 inputs of either slot,
// tools/cvtcheck.hip): one instruction where two conversions, a mask and a shift-or were four (round 5).  Spelled as the instruction for
// f2h's reason: a C cast would fuse a producing multiply into the conversion.
#if !defined(OW_CVT_PK) && defined(__gfx950__)
#define OW_CVT_PK 1
#elif !defined(OW_CVT_PK)
#define OW_CVT_PK 0
#endif
OW_DEV uint32_t f2h2(float lo, float hi);
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
OW_DEV uint32_t f2h2(float lo, float hi) {
#if OW_DEVICE_BUILD && OW_CVT_PK
    uint32_t r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
#else
    return (uint32_t)f2h(lo) | ((uint32_t)f2h(hi) << 16);
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
    int32_t fault;         // entry 0 only: fault-injection bits of this batch (ow_debug_inject_fault; 0 in normal operation)
};
// bits of the device status word (DeviceBuffers::status): set by a kernel, turned into OW_ERR_HIP by the host at the next sync
constexpr uint32_t kStatusRowSyncTimeout = 1u;  // a wave-pair rendezvous (RowSync, N = 2048) gave up waiting for its partner
constexpr unsigned long long kRowSyncTimeoutTicks = 2000000ull;  // of wall_clock64() (100 MHz constant clock): 20 ms
// layer-parallel pass 2: a block = plan_lp_rows(N) rows x 4 lane groups (one per transform) x N/16 lanes (512 threads)
constexpr int plan_lp_rows(int N) { return 128 / plan_T(N) > 0 ? 128 / plan_T(N) : 1; }
constexpr int plan_lp_threads(int N) { return plan_lp_rows(N) * kLayers * plan_T(N); }
constexpr int plan_lp_lds_cplx(int N) { return plan_region_cplx(N) * plan_lp_rows(N) * kLayers + plan_tw_total(N); }
constexpr int plan_lp_handoff_cplx(int N) { return plan_lp_threads(N) / 2; }  // tick groups, pipelined pass 2: one half's foam values for the other

// Work items of the tick-group kernel (k_tick_group_c_lp) for `slots` cascades.  Pass 2: one item = plan_lp_rows(N) rows.  Pass 1:
// one item = one BLOCK of Q side-by-side 8-row sub-items, all of the same kind, so that the block barriers inside the layer path
// stay uniform: layer 0 and layer 2 over every 8-row group, layer 1 over the groups of the upper half of the rows (the lower
// half's hz is the conjugate of the mirrored rows'), and the three extra transforms L = 3..5 of texel row 0 of every cascade.
template <int N>
struct TickPlan {
    static constexpr int Q = plan_lp_threads(N) / plan_wg_threads(N);  // pass-1 sub-items side by side in one block
    static constexpr int GPS = N / kWgRows;                            // 8-row groups per cascade
    static_assert(Q >= 1 && plan_lp_threads(N) % plan_wg_threads(N) == 0 && (GPS / 2) % Q == 0, "block shapes of the two passes must nest");
    static constexpr int full(int slots) { return slots * GPS / Q; }             // blocks of layer 0 (and of layer 2)
    static constexpr int upper(int slots) { return slots * (GPS / 2) / Q; }      // blocks of layer 1 (upper half rows)
    static constexpr int row0(int slots) { return (slots * 3 + Q - 1) / Q; }     // blocks of the three row-0 transforms
    static constexpr int items_1(int slots) { return 2 * full(slots) + upper(slots) + row0(slots); }
    // the other pass-1 item form (TickGroupArgs::p1_compact): k_pass1c's -- an 8-row group does all its layers and, for the group
    // that holds texel row 0, the three extra transforms; Q groups side by side per block, all in the same half of the rows
    // ((GPS / 2) % Q == 0), so that "skip layer 1 below N/2" stays block-uniform
    static constexpr int items_1_compact(int slots) { return full(slots); }
    static OW_HD void decode_compact(int item, int sub, int &slot, int &row0) {
        const int group = item * Q + sub;
        slot = group / GPS;
        row0 = (group % GPS) * kWgRows;
    }
    static constexpr int items_2(int slots) { return slots * (N / plan_lp_rows(N)); }
    // the pipelined pass-2 form (TickGroupArgs::p2_pipe): a block = half as many columns, its two halves work on alternate ticks
    static constexpr int kPipeRows = plan_lp_rows(N) / 2 > 0 ? plan_lp_rows(N) / 2 : 1;
    static constexpr int items_2_pipe(int slots) { return slots * (N / kPipeRows); }
    // sub-item `sub` (0..Q-1) of pass-1 item `item`: which (layer / row-0 transform L, launch slot, first row); false = this
    // sub-block has nothing to do (only in the last row-0 item)
    static OW_HD bool decode(int item, int sub, int slots, int &L, int &slot, int &row0) {
        const int n_full = full(slots), n_upper = upper(slots);
        if (item < 2 * n_full) {  // layers 0 and 2: every 8-row group
            L = item < n_full ? 0 : 2;
            const int group = (item < n_full ? item : item - n_full) * Q + sub;
            slot = group / GPS;
            row0 = (group % GPS) * kWgRows;
            return true;
        }
        if (item < 2 * n_full + n_upper) {  // layer 1: the upper half of the rows
            L = 1;
            const int group = (item - 2 * n_full) * Q + sub;
            slot = group / (GPS / 2);
            row0 = N / 2 + (group % (GPS / 2)) * kWgRows;
            return true;
        }
        const int r = (item - 2 * n_full - n_upper) * Q + sub;  // the three extra transforms of texel row 0, one (slot, L) per sub-block
        const bool active = r < slots * 3;
        slot = active ? r / 3 : 0;
        L = 3 + (active ? r % 3 : 0);
        row0 = 0;
        return active;
    }
};

// one launch of ow_run's tick groups (k_tick_group_c_lp): pass 2 of d2 consecutive ticks and pass 1 of d1 later ticks
constexpr int kMaxTickGroup = 16;  // (ticks per launch: four; eight up to 512 Ki texels per tick; twelve / sixteen for the smallest ticks -- the
                                   //  runtime's rule and its measurements: ow_runtime.hip tick_group_depth_for, profiles/r04_group_depth.txt)
struct TickGroupArgs {
    float time1[kMaxTickGroup][8];   // FP32-narrowed params.time of the d1 pass-1 ticks, per launch slot
    int32_t tbase2[kMaxTickGroup];   // first scratch slot of each pass-2 tick
    int32_t tbase1[kMaxTickGroup];   // ... of each pass-1 tick
    int32_t slots, n2, n1, d2, d1;
    int32_t pair_compact;            // k_tick_pair_c instead: the compact family's bodies, one batch of each pass --
    int32_t first2, slots2;          //   pass 2 of launch slots first2 .. first2 + slots2 - 1 (scratch slots tbase2[0] ...; slots2 = 0: none)
    int32_t first1, slots1;          //   pass 1 of launch slots first1 .. first1 + slots1 - 1 (scratch slots tbase1[0] ..., times time1[0][slot]);
                                     //   the group kernel honours first1 as well: its pass-1 items take launch slots first1 .. first1 + slots - 1
    int32_t step1;                   //   ... plus j * step1 for pass-1 tick j (group kernel only; 0: every tick the same cascades; `slots`: tick j is
                                     //   ANOTHER set of cascades at the same tick -- the cascades the reference's next ow_process calls will take)
    int32_t p2_pipe;                 // pass-2 blocks in the pipelined form (half the columns, the two halves of the block on alternate ticks)
    int32_t p1_compact;              // pass-1 items in k_pass1c's form (8 rows, all layers) instead of the layer-parallel form
};
// One launch of the TICK-PAIR kernels (k_tick_pair_c, k_tick_pair_c_split) -- pass 2 of one batch, pass 1 of the next -- needs one row of
// times, two scratch bases and the two block counts: 256 bytes with the cascades' constants instead of FrameArgs + TickGroupArgs' 950
// (round 5: the kernarg segment of the headline kernel 1 052 -> 360 bytes; same-lease A/B in profiles/r05_ab_rounds.txt).
struct PairFrame {  // CascadeFrame without the time (pass 1 takes it from PairArgs::time1, pass 2 has no use for it) and the fault word
    float tile_x, tile_y, whitecap, foam_grow_rate, foam_decay;
    int32_t cascade;
};
struct PairArgs {
    PairFrame c[8];      // per launch slot
    float time1[8];      // FP32-narrowed params.time of the pass-1 batch, per launch slot
    int32_t tbase2, tbase1;  // first scratch slot of the pass-2 / pass-1 batch
    int32_t n2, n1;          // blocks of each pass (multiples of 8; either may be 0)
    int32_t first2, first1;  // first launch slot of each batch
    int32_t fault, pad;      // fault-injection bits of this launch (tests)
};
OW_HD CascadeFrame pair_frame(const PairArgs &g, int launch_slot) {
    const PairFrame &f = g.c[launch_slot];
    CascadeFrame cf;
    cf.tile_x = f.tile_x, cf.tile_y = f.tile_y, cf.time = g.time1[launch_slot], cf.whitecap = f.whitecap;
    cf.foam_grow_rate = f.foam_grow_rate, cf.foam_decay = f.foam_decay, cf.cascade = f.cascade, cf.fault = g.fault;
    return cf;
}
// fault-injection bits (tests): kFaultRowSync = the second wave of every pair never publishes its epoch
constexpr int32_t kFaultRowSync = 1;
constexpr int kMaxCascades = 8;
struct FrameArgs {
    CascadeFrame c[kMaxCascades];
};

// ------------------------------------------------------------------------------------
// Global memory access.  Every access of the frame kernels is  (wave-uniform 128-bit buffer resource)
// + (one 32-bit per-lane byte offset) + (wave-uniform byte offset held in an SGPR): the gfx950 buffer
// instructions take all three directly, so no 64-bit per-element address is ever formed in VGPRs.
// The plain-C++ build (tests/emul) turns the same calls into pointer arithmetic.
// ------------------------------------------------------------------------------------
#if OW_DEVICE_BUILD
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
struct GBuf {
    __amdgpu_buffer_rsrc_t r;
};
OW_DEV GBuf make_gbuf(const void *base, uint32_t bytes) {
    // word 3 = 0x00020000: raw buffer, 32-bit data format (the gfx9 family default descriptor)
    return GBuf{__builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000)};
}
template <int AUX = 0>
OW_DEV f32x4 gload16(GBuf b, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b.r, (int)voff, (int)soff, AUX));
}
template <int AUX = 0>
OW_DEV cplx gload8(GBuf b, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(cplx, __builtin_amdgcn_raw_buffer_load_b64(b.r, (int)voff, (int)soff, AUX));
}
template <int AUX = 0>
OW_DEV float gload4(GBuf b, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.r, (int)voff, (int)soff, AUX));
}
template <int AUX = 0>
OW_DEV uint16_t gload2(GBuf b, uint32_t voff, uint32_t soff) {
    return (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(b.r, (int)voff, (int)soff, AUX);
}
template <int AUX = 0>
OW_DEV void gstore8(GBuf b, uint32_t voff, uint32_t soff, cplx v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v), b.r, (int)voff, (int)soff, AUX);
}
template <int AUX = 0>
OW_DEV void gstore8h(GBuf b, uint32_t voff, uint32_t soff, u16x4 v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v), b.r, (int)voff, (int)soff, AUX);
}
template <int AUX = 0>
OW_DEV void gstore8w(GBuf b, uint32_t voff, uint32_t soff, uint32_t w0, uint32_t w1) {
    __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{w0, w1}, b.r, (int)voff, (int)soff, AUX);
}
template <int AUX = 0>
OW_DEV void gstore16(GBuf b, uint32_t voff, uint32_t soff, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), b.r, (int)voff, (int)soff, AUX);
}
#else
struct GBuf {
    char *p;
};
OW_DEV GBuf make_gbuf(const void *base, uint32_t) { return GBuf{(char *)const_cast<void *>(base)}; }
template <int AUX = 0>
OW_DEV f32x4 gload16(GBuf b, uint32_t voff, uint32_t soff) { f32x4 v; __builtin_memcpy(&v, b.p + voff + soff, 16); return v; }
template <int AUX = 0>
OW_DEV cplx gload8(GBuf b, uint32_t voff, uint32_t soff) { cplx v; __builtin_memcpy(&v, b.p + voff + soff, 8); return v; }
template <int AUX = 0>
OW_DEV float gload4(GBuf b, uint32_t voff, uint32_t soff) { float v; __builtin_memcpy(&v, b.p + voff + soff, 4); return v; }
template <int AUX = 0>
OW_DEV uint16_t gload2(GBuf b, uint32_t voff, uint32_t soff) { uint16_t v; __builtin_memcpy(&v, b.p + voff + soff, 2); return v; }
template <int AUX = 0>
OW_DEV void gstore8(GBuf b, uint32_t voff, uint32_t soff, cplx v) { __builtin_memcpy(b.p + voff + soff, &v, 8); }
template <int AUX = 0>
OW_DEV void gstore8h(GBuf b, uint32_t voff, uint32_t soff, u16x4 v) { __builtin_memcpy(b.p + voff + soff, &v, 8); }
template <int AUX = 0>
OW_DEV void gstore8w(GBuf b, uint32_t voff, uint32_t soff, uint32_t w0, uint32_t w1) {
    const uint32_t w[2] = {w0, w1};
    __builtin_memcpy(b.p + voff + soff, w, 8);
}
template <int AUX = 0>
OW_DEV void gstore16(GBuf b, uint32_t voff, uint32_t soff, f32x4 v) { __builtin_memcpy(b.p + voff + soff, &v, 16); }
#endif
// cache-policy bits of the buffer instructions (aux operand): 0 = default, 2 = nt (streamed once)
// 16 = sc1: agent-scope coherence for THIS access (gfx942/950 memory model: an sc1 store writes through, an sc1 load never hits a
// line that another XCD's store could have outdated) -- what data exchanged between blocks INSIDE one kernel must use
constexpr int kAuxDefault = 0, kAuxNT = 2, kAuxAgent = 16;

// Intermediate layout (device-private), one plane per packed layer:
//   T[c][layer][y/16][x'][y%16] complex FP32 (8-byte units); one 128-byte line = 16 consecutive y of one x'.
// Pass 1 stages a layer through LDS and every wave store instruction writes 8 x' x (8 rows x 8 B = one full
// 64-byte write request); pass 2 lanes read whole 128-byte lines (16 lanes x 8 B).
OW_HD constexpr uint32_t t_unit(int n, int layer, int xp, int y) {  // index in complex units inside one cascade (< 2^24)
    return ((((uint32_t)layer * (uint32_t)(n >> 4) + (uint32_t)(y >> 4)) * (uint32_t)n + (uint32_t)xp) << 4) + (uint32_t)(y & 15);
}
constexpr uint32_t t_cascade_bytes(int n) { return (uint32_t)n * (uint32_t)n * kLayers * 8u; }

// k-vector component exactly as spectrum_modulate.glsl:60 writes it
OW_DEV float modulate_kcomp(int id, int n, float tile) {
    return ((((float)id - (float)n * 0.5f) * 2.0f) * kPi) / tile;
}

// ------------------------------------------------------------------------------------
// PASS 1, lane view.  Row y of cascade c: load h0 + omega, time-modulate (spectrum_modulate.glsl:64-70),
// then for each packed layer build the row's spectrum and run the row IFFT (fft_compute.glsl, first
// dispatch); results go to the transposed intermediate T (transpose.glsl fused into the store).
// ------------------------------------------------------------------------------------
#ifndef OW_OMEGA_ONE_LOAD
#define OW_OMEGA_ONE_LOAD 1
#endif
template <int N>
struct Pass1 {
    static constexpr int T = plan_T(N), P = kP;

    // The ifftshift sign (-1)^(x'+y') of fft_unpack.glsl:38 is not multiplied in at the end: a factor (-1)^x' on
    // the output of a length-N inverse DFT is a circular shift of its input by N/2, i.e. FFT slot j simply takes
    // the texel of slot (j + 8) % 16 (N/2 = 8*T).  Pass 1 does this along x, pass 2 along y: no instruction.
    static constexpr int rot(int j) { return (j + 8) & 15; }

    // h[j] = h(k, t) = h0 * m + conj(h0(-k)) * conj(m),  m = exp(i * omega * time)   (spectrum_modulate.glsl:64-68)
    // for the lane's texels x = t + T*rot(j) of row y.
    //
    // The reference's spectrum texel stores (h0(k), conj(h0(-k))) -- every amplitude twice.  Here only the plane
    // a[y][x] = h0(k) is kept (8 B/texel): the partner comes from the mirrored texel, zw = conj(a[(N-y)%N][(N-x)%N])
    // (spectrum_compute.glsl:121-124 evaluates exactly that), and omega is even in k, so rows beyond N/2 read
    // the mirrored row of the omega plane as well.  The block that owns the mirrored rows runs on the same XCD at
    // the same time (p1_block_to_rows), reads the same lines, and the XCD's L2 serves one of the two: HBM sees each
    // amplitude once per tick (8 + 2 B/texel instead of 16 + 4).
    // a_off / b_off: byte offsets of a[y][t] and of the mirrored texel of (t, slot 0 of the natural order);
    // b_wrap: lane 0 of a row pairs x = 0 with x = 0 (not with x = N).
    // load_raw issues the loads (a = h0(k), b = the mirrored texel, om = omega); modulate consumes them.  Apart they let a kernel
    // put other work -- the twiddle table's way into LDS and its barrier -- between issue and first use.
    // (J0 .. J1 - 1: the FFT slots whose texels are asked for.  Issuing them four at a time with the modulation of the previous four in between -- so that
    //  the arithmetic runs inside the load-issue stall of a launch's first wave generation -- was measured in round 6: nothing at 2048^2 x 4, 0.8 % slower at
    //  1024^2 x 4; profiles/r06_ab_kernel_variants.txt.)
    template <int AUX = 0, int J0 = 0, int J1 = P>
    static OW_DEV void load_raw(cplx *a, cplx *b, float *om, int t, int y, GBuf h0_c, GBuf om_c) {
        const int ym = (N - y) % N;
        const int tm = (T - t) % T;                           // lane part of the mirrored column
        const uint32_t a_off = (uint32_t)(y * N + t) * 8u;
        const uint32_t b_off = (uint32_t)(ym * N + tm) * 8u;  // + block part below
        const bool lane0 = (t == 0);
        // omega: own row for y <= N/2, else the mirrored row (same values, shared lines)
        const bool om_mirror = y > N / 2;
        const uint32_t o_off = (om_mirror ? (uint32_t)(ym * N + tm) : (uint32_t)(y * N + t)) * 4u;
#pragma unroll
        for (int j = J0; j < J1; ++j) {
            const int blk = rot(j);  // x = t + T*blk ;  mirrored x = (T - t) + T*(15 - blk)  [t > 0],  T*((16 - blk) % 16)  [t = 0]
            a[j] = gload8<AUX>(h0_c, a_off, (uint32_t)(T * blk) * 8u);
            const uint32_t mb = (uint32_t)(T * (15 - blk)), mb0 = (uint32_t)(T * ((16 - blk) % 16));
            // omega, ONE load instruction whichever half the row is in where a row is wave-uniform (N >= 1024; round 6): the two forms differ in a lane
            // offset and a wave-uniform one; written as two loads under a condition, the compiler branched around each of the sixteen, which cut the
            // load sequence into forty basic blocks.  (Where a wave holds several rows the two predicated loads stay: a per-lane select costs more.)
            constexpr bool one_load = OW_OMEGA_ONE_LOAD && T >= 64;
            if (blk == 0) {  // lane 0 needs block 0 here, the other lanes block 15: one lane-dependent offset
                b[j] = gload8<AUX>(h0_c, lane0 ? b_off : b_off + mb * 8u, 0u);
                if constexpr (one_load) om[j] = gload4<AUX>(om_c, (om_mirror && !lane0) ? o_off + mb * 4u : o_off, 0u);
                else om[j] = om_mirror ? gload4<AUX>(om_c, lane0 ? o_off : o_off + mb * 4u, 0u) : gload4<AUX>(om_c, o_off, 0u);
            } else {
                b[j] = gload8<AUX>(h0_c, lane0 ? b_off + (uint32_t)T * 8u : b_off, mb * 8u);
                (void)mb0;
                if constexpr (one_load) {
#if OW_DEVICE_BUILD
                    const bool mirror_u = __builtin_amdgcn_readfirstlane((int)om_mirror) != 0;
#else
                    const bool mirror_u = om_mirror;
#endif
                    om[j] = gload4<AUX>(om_c, (om_mirror && lane0) ? o_off + (uint32_t)T * 4u : o_off, mirror_u ? mb * 4u : (uint32_t)(T * blk) * 4u);
                } else {
                    om[j] = om_mirror ? gload4<AUX>(om_c, lane0 ? o_off + (uint32_t)T * 4u : o_off, mb * 4u)
                                      : gload4<AUX>(om_c, o_off, (uint32_t)(T * blk) * 4u);
                }
            }
        }
    }
    // TWO TEXELS PER PACKED INSTRUCTION (round 5).  The chip runs these kernels at its power limit, so every vector instruction removed comes
    // back as clock (profiles/r05_ab_rounds.txt), and a third of pass 1's instructions were per-texel SCALAR arithmetic on real coefficients --
    // the phase reduction, the wave numbers, the layer coefficients.  Texels j and j + 1 of a lane now share one packed instruction for each of
    // those steps (v_pk_mul / v_pk_fma: each half is the IEEE operation of the scalar form, so the results are bit-identical -- maps hashed against
    // the one-texel build and round 4's library, profiles/r05_hash_maps.txt; the contraction choices of the scalar form, 1 + ky / |k| as ONE fma,
    // are spelled out).  164 of k_pass1c<1024>'s 2 858 vector instructions go; measured, same box, alternating builds: 1024^2 x 4 51.66 -> 51.46 us
    // per tick, 2048^2 x 4 229.2 -> 228.8, but 512^2 x 8 27.22 -> 27.51 (and at 256 the temporaries spill) -- so from N = 1024 on only.
    // OW_P1_PAIRWISE=0 keeps the one-texel forms everywhere (A/B builds).
    static constexpr bool kPairwise = N >= 1024;
#ifndef OW_P1_PAIRWISE
#define OW_P1_PAIRWISE 7  // bits: 1 = phase reduction (modulate), 2 = wave numbers, 4 = layer coefficients
#endif
    template <int J0 = 0, int J1 = P>
    static OW_DEV void modulate(cplx *h, const cplx *a, const cplx *b, const float *om, float time) {
        static_assert(J0 % 2 == 0 && J1 % 2 == 0, "texels are modulated two at a time");
#if OW_DEVICE_BUILD && (OW_P1_PAIRWISE & 1)
        if constexpr (kPairwise) {
            // Texels j and j + 1 together, every value in "one texel per half" layout, so that nothing has to be shuffled between the steps:
            //   sincos_phase's Cody-Waite reduction  ->  its two polynomials, each for both texels  ->  the sign (-1)^n  ->  per texel
            //   h.re = fma(pp.re, c, -(pp.im s)),  h.im = fma(qq.im, c, qq.re s)   with pp = a + b, qq = a - b, (c, s) = exp(i omega t):
            // exactly the operations -- the two fused products included -- that the one-texel form below compiles to.
#pragma unroll
            for (int j = J0; j < J1; j += 2) {
                // (the FP32-rounded products omega * t of spectrum_modulate.glsl:65: only a multiply and explicit fmas consume them)
                const cplx ph = cplx{om[j], om[j + 1]} * cplx{time, time};
                const cplx q = ph * cplx{0.318309886183790672f, 0.318309886183790672f};
                const cplx n = cplx{__builtin_rintf(q.x), __builtin_rintf(q.y)};
                cplx r = __builtin_elementwise_fma(-n, cplx{3.140625f, 3.140625f}, ph);
                r = __builtin_elementwise_fma(-n, cplx{9.67502593994140625e-4f, 9.67502593994140625e-4f}, r);
                r = __builtin_elementwise_fma(-n, cplx{1.509957990978376432e-7f, 1.509957990978376432e-7f}, r);
                const cplx z = r * r;
                cplx ps = cplx{2.6343420813645935e-06f, 2.6343420813645935e-06f};
                ps = __builtin_elementwise_fma(ps, z, cplx{-0.00019822614558506757f, -0.00019822614558506757f});
                ps = __builtin_elementwise_fma(ps, z, cplx{0.008333241567015648f, 0.008333241567015648f});
                ps = __builtin_elementwise_fma(ps, z, cplx{-0.1666666567325592f, -0.1666666567325592f});
                const cplx sn = __builtin_elementwise_fma(ps * z, r, r);
                cplx pc = cplx{-2.654252000411361e-07f, -2.654252000411361e-07f};
                pc = __builtin_elementwise_fma(pc, z, cplx{2.478597525623627e-05f, 2.478597525623627e-05f});
                pc = __builtin_elementwise_fma(pc, z, cplx{-0.0013888811226934195f, -0.0013888811226934195f});
                pc = __builtin_elementwise_fma(pc, z, cplx{0.0416666679084301f, 0.0416666679084301f});
                const cplx cs = __builtin_elementwise_fma(pc * z, z, __builtin_elementwise_fma(cplx{-0.5f, -0.5f}, z, cplx{1.0f, 1.0f}));
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const uint32_t flip = (uint32_t)(int)(e ? n.y : n.x) << 31;  // (-1)^n as a sign bit
                    const float se = e ? sn.y : sn.x, ce = e ? cs.y : cs.x;
                    const float s = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, se) ^ flip);
                    const float c = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, ce) ^ flip);
                    const cplx pp = cadd(a[j + e], b[j + e]), qq = csub(a[j + e], b[j + e]);
                    h[j + e] = cplx{__builtin_fmaf(pp.x, c, -mul_rn(pp.y, s)), __builtin_fmaf(qq.y, c, mul_rn(qq.x, s))};
                    opaque_inplace(h[j + e]);
                }
                if (j % 4 == 2) OW_SCHED_FENCE();
            }
            return;
        }
#endif
#pragma unroll
        for (int j = J0; j < J1; ++j) {
            const cplx m = expi_phase(mul_rn(om[j], time));  // (cos, sin)
            // reference: h = h0 * m + conj(h0(-k)) * conj(m) with the texel (a, conj(b)), m = (cs, sn).  Expanded:
            //   h.re = (a.re + b.re) cs - (a.im + b.im) sn ,  h.im = (a.re - b.re) sn + (a.im - b.im) cs
            const cplx pp = cadd(a[j], b[j]), qq = csub(a[j], b[j]);
#if OW_DEVICE_BUILD
            const cplx t1 = pp * m, t2 = qq * m.yx;
#else
            const cplx t1 = cplx{pp.x * m.x, pp.y * m.y}, t2 = cplx{qq.x * m.y, qq.y * m.x};
#endif
            h[j] = cplx{t1.x - t1.y, t2.x + t2.y};
            opaque_inplace(h[j]);
            if (j % 4 == 3) OW_SCHED_FENCE();
        }
    }
    template <int AUX = 0>
    static OW_DEV void load_modulate(cplx *h, int t, int y, GBuf h0_c, GBuf om_c, float time) {
        cplx a[P], b[P];
        float om[P];
        load_raw<AUX>(a, b, om, t, y, h0_c, om_c);
        modulate(h, a, b, om, time);
    }

    // Wave-vector terms of the lane's 16 texels (spectrum_modulate.glsl:60-62).  kx of slot j is
    // kx0 + (T*rot(j)) * dkx with kx0 = (t - N/2) * dkx (one FMA when needed); ik[j] = 1 / (|k| + 1e-6) is kept in
    // registers for the four layers: 1/(|k| + 1e-6) = rsq(k2) * (1 - 1e-6 * rsq(k2)) + O(1e-12 / k2).  k2 carries
    // a 1e-30 bias so that the DC texel (k_vec = 0, where the reference yields k_unit = 0) stays finite.
    static OW_DEV float kx_of(int j, float kx0, float dkx) { return __builtin_fmaf((float)(T * rot(j)), dkx, kx0); }
    static OW_DEV void wave_numbers(float *ik, int t, float ky, float dkx) {
        const float kx0 = (float)(t - N / 2) * dkx, ky2 = __builtin_fmaf(ky, ky, 1e-30f);
#if OW_DEVICE_BUILD && (OW_P1_PAIRWISE & 2)
        if constexpr (kPairwise) {
#pragma unroll
            for (int j = 0; j < P; j += 2) {
                const cplx kx = kx_pair(j, kx0, dkx);
                const cplx k2 = __builtin_elementwise_fma(kx, kx, cplx{ky2, ky2});
                const cplx rk = cplx{fast_rsq(k2.x), fast_rsq(k2.y)};
                const cplx ikp = __builtin_elementwise_fma(cplx{-1e-6f, -1e-6f} * rk, rk, rk);
                ik[j] = ikp.x;
                ik[j + 1] = ikp.y;
            }
            return;
        }
#endif
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const float kx = kx_of(j, kx0, dkx);
            const float rk = fast_rsq(__builtin_fmaf(kx, kx, ky2));
            ik[j] = __builtin_fmaf(-1e-6f * rk, rk, rk);
        }
    }
#if OW_DEVICE_BUILD
    // kx of texels j and j + 1 (kx_of twice, one packed fma)
    static OW_DEV cplx kx_pair(int j, float kx0, float dkx) {
        return __builtin_elementwise_fma(cplx{(float)(T * rot(j)), (float)(T * rot(j + 1))}, cplx{dkx, dkx}, cplx{kx0, kx0});
    }
#endif

    // d[j] = packed layer L at texel x (spectrum_modulate.glsl:72-89); each layer is h times a complex
    // coefficient of the wave vector (u = k / |k|):  L0 = i(1+uy) h, L1 = (-ky + i ux) h, L2 = i(kx - ky uy) h,
    // L3 = -ux (kx + i ky) h; written as  c*h + e*(i h)  so that each is two or three packed instructions.
    struct NoHook {
        OW_DEV void operator()(int) const {}
    };
    // after_group(g) is called after texel group g = 0..3 (the pass-1 kernel drains the previous layer's staged
    // rows there, a few stores at a time, instead of in one burst)
    template <int L, class Hook = NoHook>
    static OW_DEV void layer_input(cplx *d, const cplx *h, const float *ik, int t, float ky, float dkx, Hook after_group = Hook()) {
        const float kx0 = (float)(t - N / 2) * dkx;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const cplx ih = cmuli(h[j]);
            if (L == 0) d[j] = cscale(ih, 1.0f + ky * ik[j]);
            if (L == 1) d[j] = cadd(cscale(h[j], -ky), cscale(ih, kx_of(j, kx0, dkx) * ik[j]));
            if (L == 2) {
                const float kx = kx_of(j, kx0, dkx);
                d[j] = cscale(ih, __builtin_fmaf(-ky, ky * ik[j], kx));
            }
            if (L == 3) {
                const float kx = kx_of(j, kx0, dkx), mux = -(kx * ik[j]);
                d[j] = cadd(cscale(h[j], mux * kx), cscale(ih, mux * ky));
            }
            opaque_inplace(d[j]);  // pins this texel's arithmetic here (pure ops would otherwise sink to their first use)
            if (j % 4 == 3) {
                OW_SCHED_FENCE();  // four texels' temporaries at a time, not sixteen
                after_group(j / 4);
                OW_SCHED_FENCE();
            }
        }
    }

    // ---- compact intermediate: three layers instead of four (tests/test_compact_math.py holds the algebra) ----
    // Off the two Nyquist lines all eight fields of spectrum_modulate.glsl:72-82 are real and three of them are
    // i*ky times three others (dhx_dx = i ky hx, dhy_dx = i ky hy, dhz_dx = i ky hz), ky being the axis PASS 2
    // transforms: pass 2 can form them itself from what it loads.  Only five real fields cross the intermediate:
    //   C0 = hx + i hy = i (1 + uy) h      C1 = hz = i ux h  (alone)      C2 = dhy_dz + i dhz_dz = i kx (1 - ux) h
    // hz travels alone because its spectrum is Hermitian along ky: pass 2's transform of (1 - ky) hz = hz + i (i ky hz) is
    // hz + i dhz_dx, two real fields again -- and only the rows with ky >= 0 (y >= N/2) have to be transformed and stored
    // at all: row N - y is the conjugate (Pass2::load_c1).  Blocks of the lower half skip layer 1.
    // On texel column id.x = 0 (lane 0, slot kColSlot; kx = -N/2 dkx is not mirrored there) the
    // reference's layers are not Hermitian-consistent; what they leak is reproduced in closed form:
    //   C1 <- 0,   C2 <- ux (ky - i kx) h,   and pass 2 adds  column_term = (kx + i ux) h  to its derived  i ky C0.
    // Texel row id.y = 0 is carried separately as three extra transforms (row0_input).
    static constexpr int kCompactLayers = 3, kColSlot = 8;  // rot(kColSlot) == 0
    template <int L, class Hook = NoHook>
    static OW_DEV void layer_input_c(cplx *d, const cplx *h, const float *ik, int t, float ky, float dkx, Hook after_group = Hook()) {
        const float kx0 = (float)(t - N / 2) * dkx;
#if OW_DEVICE_BUILD && (OW_P1_PAIRWISE & 4)
        static_assert(kColSlot % 2 == 0, "the pair (kColSlot, kColSlot + 1) is computed at kColSlot");
        cplx kx2 = cplx{0.0f, 0.0f}, ux2 = kx2, cf2 = kx2;  // kx, ux = kx / |k| and the layer's real coefficient of i h, texels j (low half) and j + 1
#endif
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const cplx ih = cmuli(h[j]);
            float kx, ux, coef;
#if OW_DEVICE_BUILD && (OW_P1_PAIRWISE & 4)
            if constexpr (kPairwise) {
                if (j % 2 == 0) {  // the pair's arithmetic in packed instructions; the odd texel takes the upper halves
                    kx2 = kx_pair(j, kx0, dkx);
                    const cplx ikp = cplx{ik[j], ik[j + 1]};
                    ux2 = kx2 * ikp;
                    cf2 = L == 0   ? __builtin_elementwise_fma(cplx{ky, ky}, ikp, cplx{1.0f, 1.0f})  // 1 + ky / |k|: ONE fma, as the one-texel form contracts
                          : L == 1 ? ux2
                                   : __builtin_elementwise_fma(-kx2, ux2, kx2);
                }
                kx = kx2.x, ux = ux2.x;  // (read at j == kColSlot only: an even slot)
                coef = (j % 2 == 0) ? cf2.x : cf2.y;
            } else
#endif
            {
                kx = kx_of(j, kx0, dkx), ux = kx * ik[j];
                coef = L == 0 ? __builtin_fmaf(ky, ik[j], 1.0f) : L == 1 ? ux : __builtin_fmaf(-kx, ux, kx);
            }
            d[j] = cscale(ih, coef);
            if (j == kColSlot && L > 0) {
                const cplx line = (L == 1) ? cplx{0.0f, 0.0f} : cadd(cscale(h[j], ux * ky), cscale(ih, -(ux * kx)));
                d[j] = cplx{t == 0 ? line.x : d[j].x, t == 0 ? line.y : d[j].y};
            }
            opaque_inplace(d[j]);
            if (j % 4 == 3) {
                OW_SCHED_FENCE();
                after_group(j / 4);
                OW_SCHED_FENCE();
            }
        }
    }
    // (kx + i ux) h at the lane's texel of slot kColSlot: meaningful in lane 0 (texel x = 0) only
    static OW_DEV cplx column_term(const cplx *h, const float *ik, int t, float dkx) {
        const float kx = kx_of(kColSlot, (float)(t - N / 2) * dkx, dkx), ux = kx * ik[kColSlot];
        return cadd(cscale(h[kColSlot], kx), cscale(cmuli(h[kColSlot]), ux));
    }
    // Texel row id.y = 0 (ky = -N/2 dky): the pass-2 inputs at ky-index 0 are the row transforms of
    //   Q1 = -ky uy h  (-> dhx_dx + i dhy_dx)     Q2 = (i ux - ky) h  (-> hz + i dhz_dx)
    //   Q3 = (i kx (1 - ux) + ky ux) h  (-> dhy_dz + i dhz_dz);   Q0 = C0 is the row's ordinary layer 0.
    // The corner texel (0, 0) mirrors onto itself and has its own forms.
    template <int Q>
    static OW_DEV void row0_input(cplx *d, const cplx *h, const float *ik, int t, float ky, float dkx) {
        const float kx0 = (float)(t - N / 2) * dkx;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const cplx ih = cmuli(h[j]);
            const float kx = kx_of(j, kx0, dkx), ux = kx * ik[j], uy = ky * ik[j];
            const bool corner = (j == kColSlot) && (t == 0);
            float a = 0.0f, b = 0.0f;  // d = a h + b (i h)
            if (Q == 1) { a = corner ? kx - ky * uy : -(ky * uy); b = corner ? ux : 0.0f; }
            if (Q == 2) { a = -ky; b = corner ? -(ky * ux) : ux; }
            if (Q == 3) { a = corner ? 0.0f : ky * ux; b = corner ? -(kx * ux) : __builtin_fmaf(-kx, ux, kx); }
            d[j] = cadd(cscale(h[j], a), cscale(ih, b));
            opaque_inplace(d[j]);
            if (j % 4 == 3) OW_SCHED_FENCE();
        }
    }

    // Transposed store of one layer.  stage_write: every lane puts its own row's results, x'-ordered, into
    // its row region (8 B per x', linear).  After a workgroup (LDS) barrier, stage_store: thread tau of the
    // block takes (row q = tau % 8, x' = tau / 8 + T*k), so that 8 consecutive lanes write the 8 consecutive
    // 8-byte units (= 64 B) of one x'.  A second barrier must follow before the regions are written again.
    static OW_DEV void stage_write(const cplx *d, int t, cplx *lds_row) {
#pragma unroll
        for (int o = 0; o < P; ++o) lds_row[t + T * o] = d[OutMap<N>::slot_of(o)];
    }
    // tau = thread index in the block, row0 = first map row of the block (multiple of 8), T_c = this cascade's T
    template <int AUX>
    static OW_DEV void stage_store(int tau, int layer, int row0, const cplx *lds_block, GBuf T_c) {
        const int q = tau % kWgRows, xi = tau / kWgRows;  // xi in [0, T)
        const cplx *st = lds_block + q * plan_region_cplx(N);
        const uint32_t voff = t_unit(N, 0, xi, row0 + q) * 8u;
        cplx v[P];
#pragma unroll
        for (int k = 0; k < P; ++k) v[k] = lds_read(st + xi + T * k);
#pragma unroll
        for (int k = 0; k < P; ++k) gstore8<AUX>(T_c, voff, (t_unit(N, layer, 0, 0) + t_unit(N, 0, T * k, 0)) * 8u, v[k]);
    }
    // the same, quarter g = 0..3 of the x' range only (k = 4g .. 4g+3)
    template <int AUX>
    static OW_DEV void stage_store_chunk(int tau, int layer, int row0, const cplx *lds_block, GBuf T_c, int g) {
        const int q = tau % kWgRows, xi = tau / kWgRows;
        const cplx *st = lds_block + q * plan_region_cplx(N);
        const uint32_t voff = t_unit(N, 0, xi, row0 + q) * 8u;
        cplx v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = lds_read(st + xi + T * (4 * g + k));
#pragma unroll
        for (int k = 0; k < 4; ++k) gstore8<AUX>(T_c, voff, (t_unit(N, layer, 0, 0) + t_unit(N, 0, T * (4 * g + k), 0)) * 8u, v[k]);
    }
};

// ------------------------------------------------------------------------------------
// PASS 2, lane view.  Row x' of T: row IFFT of the 4 layers (fft_compute.glsl, second dispatch), then
// fft_unpack.glsl:38-68 fused: ifftshift sign, displacement, Jacobian/foam RMW, normal.  The layers are
// taken in the order 2, 3, 1, 0 so that at most two layers' worth of results wait in registers:
//   after 2: dhy_dz, dhx_dx            after 3: Jacobian -> foam, gy  (kept as packed halves) + dhx_dx
//   after 1: gx -> normal map store; hz kept      after 0: displacement store
// ------------------------------------------------------------------------------------
template <int N>
struct Pass2 {
    static constexpr int T = plan_T(N), P = kP;

    // byte offset (lane part / uniform part) of T[layer][x'][y = t + T*j] inside the cascade
    static OW_DEV uint32_t t_voff(int t, int xp) {
        return (T >= 16 ? t_unit(N, 0, xp, t) : t_unit(N, 0, xp, 0) + (uint32_t)t) * 8u;
    }
    static constexpr uint32_t t_soff(int layer, int j) {
        return (t_unit(N, layer, 0, 0) + (T >= 16 ? t_unit(N, 0, 0, T * j) : t_unit(N, 0, 0, 16 * (j / 2)) + 8u * (j % 2))) * 8u;
    }
    // one packed layer of row x': d[j] = T[layer][x'][y = t + T*rot(j)]  (rot: the (-1)^y' half of the ifftshift sign,
    // see Pass1; T already carries the (-1)^x' half)
    static constexpr int rot(int j) { return (j + 8) & 15; }
    template <int AUX>
    static OW_DEV void load_layer(cplx *d, int t, int xp, int layer, GBuf T_c) {
        const uint32_t voff = t_voff(t, xp);
#pragma unroll
        for (int j = 0; j < P; ++j) d[j] = gload8<AUX>(T_c, voff, t_soff(layer, rot(j)));
    }
    // Foam state.  The reference re-reads normal_map.a (fft_unpack.glsl:61); reading 2 useful bytes out of every
    // 8-byte texel would pull the whole normal map back in, so the context keeps a private FP16 copy of the foam
    // channel in the order this kernel consumes it: foam[x'][t][o] = foam of texel (row x', col t + T*o), i.e. one
    // lane's 16 values are 32 contiguous bytes (two 16-byte accesses instead of sixteen 2-byte ones, 2 B/texel
    // read + 2 B/texel written instead of 8 B/texel read).  Same FP16 bits as normal.a, so the recurrence is
    // unchanged.  pk[i] holds the halves of o = 2i (low) and 2i + 1 (high).
    OW_HD static constexpr uint32_t foam_index(int xp, int yp) { return (uint32_t)xp * N + (uint32_t)(yp % T) * 16u + (uint32_t)(yp / T); }
    static OW_DEV void load_foam(uint32_t *pk, int t, int xp, GBuf foam_c) {
        const uint32_t voff = foam_index(xp, t) * 2u;
        const f32x4 a = gload16(foam_c, voff, 0u), b = gload16(foam_c, voff, 16u);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) pk[i] = __builtin_bit_cast(uint32_t, v[i]);
    }
    static OW_DEV void store_foam(const uint32_t *pk, int t, int xp, GBuf foam_c) {
        const uint32_t voff = foam_index(xp, t) * 2u;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_bit_cast(float, pk[i]);
        gstore16(foam_c, voff, 0u, f32x4{v[0], v[1], v[2], v[3]});
        gstore16(foam_c, voff, 16u, f32x4{v[4], v[5], v[6], v[7]});
    }

    // optional FP32 debug image: 8 pre-quantisation channels per texel [hx,hy,hz,gx,gy,dhx_dx,foam,J]
    static OW_DEV void f32_put(GBuf f32_c, uint32_t tex, int o, int ch, float v) {
#if OW_DEVICE_BUILD
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), f32_c.r, (int)(tex * 32u + (uint32_t)ch * 4u), (int)((uint32_t)(T * o) * 32u), 0);
#else
        __builtin_memcpy(f32_c.p + tex * 32u + (uint32_t)ch * 4u + (uint32_t)(T * o) * 32u, &v, 4);
#endif
    }

    // layer 2 done (fft_unpack.glsl:54): dhy_dz = re, dhx_dx = im of output ordinal o: OutMap<N>::slot_of(o) of d
    // (nothing to compute: the caller keeps the layer's registers)
    // layer 3 done (fft_unpack.glsl:55-66): Jacobian, foam RMW, gy; gy_foam[o] = packed halves (gy | foam << 16)
    // foam_pk: in = previous foam halves, out = new foam halves (see load_foam)
    template <bool F32>
    static OW_DEV void after_layer3(const cplx *l3, const cplx *l2, uint32_t *foam_pk, uint32_t *gy_foam,
                                    uint32_t tex, const CascadeFrame &cf, GBuf f32_c) {
#pragma unroll
        for (int o = 0; o < P; ++o) {
            const int sl = OutMap<N>::slot_of(o);
            const float dhy_dz = l2[sl].x, dhx_dx = l2[sl].y;
            const float dhz_dz = l3[sl].x, dhz_dx = l3[sl].y;
            const float jac = (1.0f + dhx_dx) * (1.0f + dhz_dz) - dhz_dx * dhz_dx;
            const float foam_factor = -fminf(0.0f, jac - cf.whitecap);
            float foam = h2f((uint16_t)((foam_pk[o / 2] >> (16 * (o & 1))) & 0xFFFFu));
            foam = mul_rn(foam, cf.foam_decay);
            foam = foam + mul_rn(foam_factor, cf.foam_grow_rate);
            foam = fminf(fmaxf(foam, 0.0f), 1.0f);
            // 1-ulp reciprocal: far inside the FP16 output step
            const float gy = dhy_dz * fast_rcp(1.0f + fabsf(dhz_dz));
            const uint32_t foam_h = f2h(foam);
            gy_foam[o] = (uint32_t)f2h(gy) | (foam_h << 16);
            foam_pk[o / 2] = (o & 1) ? ((foam_pk[o / 2] & 0xFFFFu) | (foam_h << 16)) : ((foam_pk[o / 2] & 0xFFFF0000u) | foam_h);
            if (F32) {
                f32_put(f32_c, tex, o, 4, gy);
                f32_put(f32_c, tex, o, 6, foam);
                f32_put(f32_c, tex, o, 7, jac);
            }
        }
    }
    // layer 1 done (fft_unpack.glsl:45,53,65-67): gx = dhy_dx / (1 + |dhx_dx|), normal map store; hz = l1.re stays in l1
    template <bool F32, int AUX>
    static OW_DEV void after_layer1(const cplx *l1, const float *dhx_dx, const uint32_t *gy_foam, uint32_t tex, GBuf norm_c,
                                    GBuf f32_c) {
#pragma unroll
        for (int o = 0; o < P; ++o) {
            const int sl = OutMap<N>::slot_of(o);
            const float gx = l1[sl].y * fast_rcp(1.0f + fabsf(dhx_dx[o]));
            gstore8h<AUX>(norm_c, tex * 8u, (uint32_t)(T * o) * 8u,
                          u16x4{f2h(gx), (uint16_t)(gy_foam[o] & 0xFFFFu), f2h(dhx_dx[o]), (uint16_t)(gy_foam[o] >> 16)});
            if (F32) {
                f32_put(f32_c, tex, o, 3, gx);
                f32_put(f32_c, tex, o, 5, dhx_dx[o]);
            }
        }
    }
    // ---- compact intermediate (see Pass1::layer_input_c): four transforms from three loaded layers ----
    //   F2 = row transform of (1 - ky) C1 -> (hz, dhz_dx)             F0 = of C0 -> (hx, hy)
    //   F1 = of  i ky C0 + (-1)^x' P  -> (dhx_dx, dhy_dx)              F3 = of C2 -> (dhy_dz, dhz_dz)
    // with element ky-index 0 (lane 0, slot kRow0Slot) of F1..F3 replaced by the separately transformed texel row 0.
    static constexpr int kRow0Slot = 8;  // rot(kRow0Slot) == 0
    static OW_DEV float ky_of(int j, int t, float dky) { return (float)(t + T * (rot(j) - 8)) * dky; }
    // P(ky): one complex per map row y, stored in this kernel's lane order (a lane's 16 values = 128 contiguous bytes)
    OW_HD static constexpr uint32_t pcol_index(int y) { return (uint32_t)(y % T) * 16u + (uint32_t)(y / T); }
    // d[j] = i ky d[j] + sign * P[y_j]; slots j < 8 take the natural-order blocks o = 8..15 of P, slots j >= 8 blocks 0..7: two
    // halves, so that only eight of P's values are in registers at a time
    template <int AUX = 0>
    static OW_DEV void derive_dx(cplx *d, int t, int xp, float dky, GBuf pcol_c) {
        const float s = (xp & 1) ? -1.0f : 1.0f;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            cplx p[P / 2];  // p[k] = P block o = 8 * (1 - half) + k
#pragma unroll
            for (int i = 0; i < P / 4; ++i) {
                const f32x4 v = gload16<AUX>(pcol_c, (uint32_t)t * 128u, 16u * (uint32_t)(i + (P / 4) * (1 - half)));
                p[2 * i] = cplx{v.x, v.y};
                p[2 * i + 1] = cplx{v.z, v.w};
            }
#pragma unroll
            for (int k = 0; k < P / 2; ++k) {
                const int j = 8 * half + k;  // rot(j) = 8 * (1 - half) + k
                const float ky = ky_of(j, t, dky);
                d[j] = cplx{__builtin_fmaf(-ky, d[j].y, s * p[k].x), __builtin_fmaf(ky, d[j].x, s * p[k].y)};
            }
            OW_SCHED_FENCE();
        }
    }
    // C1 = (1 - ky) hz for all y of row x' from the stored half S(x', y >= N/2) = row transform of hz:
    //   slots j < 8  (y = t + T (8 + j) >= N/2):  (1 - ky) S(y)
    //   slots j >= 8 (y = t + T (j - 8) <  N/2):  (1 - ky) conj(S(N - y)),  N - y = (N - t) - T (j - 8): the lanes of the wave read
    //   the same lines in reverse order (L1 hits); slot 8 of lane 0 (y = 0) reads nonsense and is replaced by the row-0 entry.
    template <int AUX>
    static OW_DEV void load_c1(cplx *d, int t, int xp, float dky, GBuf T_c) {
        static_assert(T >= 16, "the mirrored offsets below assume that T is a multiple of the 16-row line");
        const uint32_t voff = t_voff(t, xp);
        // lane part of the mirrored address for the smallest mirrored y the lane needs (j = 15): (N - t) - 7 T
        const uint32_t voff_m = t_unit(N, 0, xp, N - t - 7 * T) * 8u;
        const float kyb = (float)t * dky;
        // all sixteen loads first, the arithmetic after them: written as one loop the compiler has been seen to wait for each load in
        // turn (sixteen dependent round trips) when this is inlined into the pipelined tick-group loop
        cplx v[P];
#pragma unroll
        for (int j = 0; j < P; ++j)
            v[j] = j < P / 2 ? gload8<AUX>(T_c, voff, t_soff(1, rot(j))) : gload8<AUX>(T_c, voff_m, (t_unit(N, 1, 0, 0) + t_unit(N, 0, 0, T * (15 - j))) * 8u);
        OW_SCHED_FENCE();
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const float sc = 1.0f - __builtin_fmaf((float)(T * (rot(j) - 8)), dky, kyb);  // 1 - ky(y_j)
            d[j] = j < P / 2 ? cscale(v[j], sc) : cplx{v[j].x * sc, v[j].y * -sc};
        }
    }
    static OW_DEV void put_row0(cplx *d, int t, cplx r) {
        d[kRow0Slot] = cplx{t == 0 ? r.x : d[kRow0Slot].x, t == 0 ? r.y : d[kRow0Slot].y};
    }
    // Order of the four transforms: F2, F0, F1, F3.  C0 feeds both F0 and F1; it is loaded ONCE and a copy stays in registers
    // across F0's transform (re-reading it two transforms later would miss the L2 and put the fourth layer's bytes back on
    // the memory interface).  What waits in registers between the phases:
    //   after F2: hz as halves, two per word (8) + c2 = dhz_dx^2 (16)         [+ the copy of C0 (32) during F0]
    //   after F0: displacement stored; c2                                      after F1: c2, dhx_dx (16), gx halves (8)
    //   after F3: Jacobian = (1 + dhx_dx)(1 + dhz_dz) - c2 -> foam, gy, normal map store
    template <bool F32>
    static OW_DEV void after_f2(const cplx *f2, uint32_t *hz_pk, float *c2, uint32_t tex, GBuf f32_c) {
#pragma unroll
        for (int o = 0; o < P; o += 2) {
            const int sl = OutMap<N>::slot_of(o), sl1 = OutMap<N>::slot_of(o + 1);
            c2[o] = f2[sl].y * f2[sl].y;
            c2[o + 1] = f2[sl1].y * f2[sl1].y;
            // packed HERE (pure ops would otherwise sink to the store, leaving both floats live across two transforms)
            hz_pk[o / 2] = (uint32_t)opaque((int)f2h2(f2[sl].x, f2[sl1].x));
            if (F32) {
                f32_put(f32_c, tex, o, 2, f2[sl].x);
                f32_put(f32_c, tex, o + 1, 2, f2[sl1].x);
            }
        }
    }
    template <bool F32, int AUX>
    static OW_DEV void after_f0(const cplx *f0, const uint32_t *hz_pk, int t, int xp, uint32_t tex, GBuf disp_c, GBuf f32_c) {
        const uint32_t w = (uint32_t)((xp ^ t) & 1) << 31;  // see after_layer0: the sign of .w's zero, as the upper half of the texel's second word
#pragma unroll
        for (int o = 0; o < P; ++o) {
            const int sl = OutMap<N>::slot_of(o);
            const uint32_t hz_h = (hz_pk[o / 2] >> (16 * (o & 1))) & 0xFFFFu;
            gstore8w<AUX>(disp_c, tex * 8u, (uint32_t)(T * o) * 8u, f2h2(f0[sl].x, f0[sl].y), hz_h | w);
            if (F32) {
                f32_put(f32_c, tex, o, 0, f0[sl].x);
                f32_put(f32_c, tex, o, 1, f0[sl].y);
            }
        }
    }
    template <bool F32>
    static OW_DEV void after_f1(const cplx *f1, float *dhx_dx, uint32_t *gx_pk, uint32_t tex, GBuf f32_c) {
#pragma unroll
        for (int o = 0; o < P; o += 2) {
            const int sl = OutMap<N>::slot_of(o), sl1 = OutMap<N>::slot_of(o + 1);
            dhx_dx[o] = f1[sl].x;
            dhx_dx[o + 1] = f1[sl1].x;
            const float gx = f1[sl].y * fast_rcp(1.0f + fabsf(dhx_dx[o])), gx1 = f1[sl1].y * fast_rcp(1.0f + fabsf(dhx_dx[o + 1]));
            gx_pk[o / 2] = (uint32_t)opaque((int)f2h2(gx, gx1));
            if (F32) {
                f32_put(f32_c, tex, o, 3, gx);
                f32_put(f32_c, tex, o, 5, dhx_dx[o]);
                f32_put(f32_c, tex, o + 1, 3, gx1);
                f32_put(f32_c, tex, o + 1, 5, dhx_dx[o + 1]);
            }
        }
    }
    // fft_unpack.glsl:55-67 with everything at hand
    template <bool F32, int AUX>
    static OW_DEV void after_f3(const cplx *f3, const float *dhx_dx, const float *c2, const uint32_t *gx_pk, uint32_t *foam_pk,
                                uint32_t tex, const CascadeFrame &cf, GBuf norm_c, GBuf f32_c) {
#pragma unroll
        for (int o = 0; o < P; o += 2) {  // two texels at a time: one word of foam_pk / gx_pk
            float foam[2], gy[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int sl = OutMap<N>::slot_of(o + e);
                const float dhy_dz = f3[sl].x, dhz_dz = f3[sl].y;
                const float jac = (1.0f + dhx_dx[o + e]) * (1.0f + dhz_dz) - c2[o + e];
                const float foam_factor = -fminf(0.0f, jac - cf.whitecap);
                float fm = h2f((uint16_t)((foam_pk[o / 2] >> (16 * e)) & 0xFFFFu));
                fm = mul_rn(fm, cf.foam_decay);
                fm = fm + mul_rn(foam_factor, cf.foam_grow_rate);
                foam[e] = fminf(fmaxf(fm, 0.0f), 1.0f);
                gy[e] = dhy_dz * fast_rcp(1.0f + fabsf(dhz_dz));
                if (F32) {
                    f32_put(f32_c, tex, o + e, 4, gy[e]);
                    f32_put(f32_c, tex, o + e, 6, foam[e]);
                    f32_put(f32_c, tex, o + e, 7, jac);
                }
            }
            foam_pk[o / 2] = f2h2(foam[0], foam[1]);
            const uint32_t gy_pk = f2h2(gy[0], gy[1]), gxw = gx_pk[o / 2];
            // normal texel = (gx, gy | dhx_dx, foam) as two words
            gstore8w<AUX>(norm_c, tex * 8u, (uint32_t)(T * o) * 8u, (gxw & 0xFFFFu) | (gy_pk << 16), f2h2(dhx_dx[o], foam[0]));
            gstore8w<AUX>(norm_c, tex * 8u, (uint32_t)(T * (o + 1)) * 8u, (gxw >> 16) | (gy_pk & 0xFFFF0000u), f2h2(dhx_dx[o + 1], foam[1]));
        }
    }

    // The whole of fft_unpack.glsl:44-67 for ONE texel whose four layer values are at hand (layer-parallel pass 2).
    // foam_prev / foam_new: FP16 bits of the recurrent state.  o = the lane's output ordinal (texel t + T*o).
    template <bool F32, int AUX>
    static OW_DEV uint16_t unpack_texel(cplx l0, cplx l1, cplx l2, cplx l3, uint16_t foam_prev, int t, int xp, int o, uint32_t tex,
                                        const CascadeFrame &cf, GBuf disp_c, GBuf norm_c, GBuf f32_c) {
        const float dhy_dz = l2.x, dhx_dx = l2.y, dhz_dz = l3.x, dhz_dx = l3.y;
        const float jac = (1.0f + dhx_dx) * (1.0f + dhz_dz) - dhz_dx * dhz_dx;
        const float foam_factor = -fminf(0.0f, jac - cf.whitecap);
        float foam = h2f(foam_prev);
        foam = mul_rn(foam, cf.foam_decay);
        foam = foam + mul_rn(foam_factor, cf.foam_grow_rate);
        foam = fminf(fmaxf(foam, 0.0f), 1.0f);
        const float gy = dhy_dz * fast_rcp(1.0f + fabsf(dhz_dz));
        const float gx = l1.y * fast_rcp(1.0f + fabsf(dhx_dx));
        const uint32_t n1 = f2h2(dhx_dx, foam);
        const uint16_t foam_h = (uint16_t)(n1 >> 16);
        const uint32_t w = (uint32_t)((xp ^ t) & 1) << 31;
        gstore8w<AUX>(norm_c, tex * 8u, (uint32_t)(T * o) * 8u, f2h2(gx, gy), n1);
        gstore8w<AUX>(disp_c, tex * 8u, (uint32_t)(T * o) * 8u, f2h2(l0.x, l0.y), (uint32_t)f2h(l1.x) | w);
        if (F32) {
            f32_put(f32_c, tex, o, 0, l0.x);
            f32_put(f32_c, tex, o, 1, l0.y);
            f32_put(f32_c, tex, o, 2, l1.x);
            f32_put(f32_c, tex, o, 3, gx);
            f32_put(f32_c, tex, o, 4, gy);
            f32_put(f32_c, tex, o, 5, dhx_dx);
            f32_put(f32_c, tex, o, 6, foam);
            f32_put(f32_c, tex, o, 7, jac);
        }
        return foam_h;
    }

    // layer 0 done (fft_unpack.glsl:44-50): displacement = (hx, hy, hz, 0) * sign; the sign only survives in the
    // zero of .w (0 * -1 = -0): sign bit = parity of x' + y', and y' = t + T*o has the parity of t
    template <bool F32, int AUX>
    static OW_DEV void after_layer0(const cplx *l0, const float *hz, int t, int xp, uint32_t tex, GBuf disp_c, GBuf f32_c) {
        const uint16_t w = (uint16_t)(((xp ^ t) & 1) << 15);
#pragma unroll
        for (int o = 0; o < P; ++o) {
            const int sl = OutMap<N>::slot_of(o);
            gstore8h<AUX>(disp_c, tex * 8u, (uint32_t)(T * o) * 8u, u16x4{f2h(l0[sl].x), f2h(l0[sl].y), f2h(hz[o]), w});
            if (F32) {
                f32_put(f32_c, tex, o, 0, l0[sl].x);
                f32_put(f32_c, tex, o, 1, l0[sl].y);
                f32_put(f32_c, tex, o, 2, hz[o]);
            }
        }
    }
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
    // correctly rounded tanhf -- which is exactly 1.0f from a = 9.02 on (1 - tanh a = 2 / (e^2a + 1) < 2^-25 there): the FP64 evaluation only for the
    // few texels around DC that need it (a wave-uniform skip almost everywhere)
    const float b = a > 9.1f ? 1.0f : (float)tanh((double)a);
    return sqrtf(kG * k * b);
}

OW_DEV void hash_uniform(uint32_t x, uint32_t y, float &u1, float &u2) {  // spectrum_compute.glsl:34-41
    uint32_t h32 = y + 374761393u + x * 3266489917u;
    h32 = 2246822519u * (h32 ^ (h32 >> 15));
    h32 = 3266489917u * (h32 ^ (h32 >> 13));
    const uint32_t n = h32 ^ (h32 >> 16), n2 = n * 48271u;
    const float rden = 1.0f / 2147483648.0f;  // 1 / float(0x7FFFFFFF): float(0x7FFFFFFF) rounds to 2^31, and a division by a power of two IS this product
    u1 = (float)((n >> 1) & 0x7FFFFFFFu) * rden;
    u2 = (float)((n2 >> 1) & 0x7FFFFFFFu) * rden;
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

// THE SAME AMPLITUDE AT HALF THE INSTRUCTIONS (round 6; the kernel's form -- the literal one above stays the CPU build's, bit-equal to the oracle).
// k_spectrum is pure arithmetic (2 000 vector instructions per texel, 56 us per 1024^2 cascade: the only kernel of the path at 3 % of the roofline), and
// most of it is generality the formulas do not need: five powf -- two with the integer exponents 5 and 4, one with the constant base 3.3, two with positive
// bases, for which exp2(e log2 x) on the hardware's 1-ulp v_exp_f32 / v_log_f32 is exact to ~1e-6 at the exponents that occur --, sinf / cosf with their
// huge-argument paths for the Box-Muller angle in [0, 2 pi) (the frame kernels' Cody-Waite sincos: 1.3e-7 absolute, on a factor of the amplitude), IEEE division
// sequences where 2-ulp reciprocals do, a division by 2^31, and tanh(k depth), which IS 1.0f from k depth = 9.02 on (1 - tanh a < 2^-25), i.e. for all
// but a few texels around DC at the reference's depth of 20 m.  h0 stays within the 2e-5 the parity tests allow against the oracle (measured on the CPU
// build, tests/test_emul.py: <= 3e-6 over the presets; the GPU's native exp2 / log2 add an ulp each).  omega is untouched (omega_texel: bit-exact).
OW_DEV float fast_exp2(float x) {
#if OW_DEVICE_BUILD
    return __builtin_amdgcn_exp2f(x);
#else
    return exp2f(x);
#endif
}
OW_DEV float fast_log2(float x) {
#if OW_DEVICE_BUILD
    return __builtin_amdgcn_logf(x);  // v_log_f32: log2
#else
    return log2f(x);
#endif
}
OW_DEV float pow_pos(float x, float e) {  // x >= 0; x^0 = 1 (also for x = 0, as powf)
    return e == 0.0f ? 1.0f : fast_exp2(e * fast_log2(x));
}
OW_DEV float tanh_sat(float a) {  // tanhf(a) for a >= 0: exactly 1 where the correctly rounded value is
    return a > 9.1f ? 1.0f : tanhf(a);
}
OW_DEV cplx spectrum_amplitude_fast(int idx, int idy, int n, const SpectrumPC &pc) {  // spectrum_compute.glsl:103-115, operation for operation up to the items above
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const float dkx = (2.0f * kPi) * fast_rcp(pc.tile_x), dky = (2.0f * kPi) * fast_rcp(pc.tile_y);
    const float half = (float)n * 0.5f;
    const float kx = ((float)idx - half) * dkx, ky = ((float)idy - half) * dky;
    const float k = sqrtf(kx * kx + ky * ky) + 1e-6f;
    const float theta = atan2f(kx, ky);
    // dispersion_relation (:58-66)
    const float a = k * pc.depth, b = tanh_sat(a);
    const float w = sqrtf(kG * k * b), rw = fast_rcp(w);
    const float dw = (0.5f * kG) * (b + a * (1.0f - b * b)) * rw;
    const float w_norm = dw * fast_rcp(k) * dkx * dky;
    // TMA_spectrum (:89-101)
    const float w_p = pc.peak_frequency;
    const float sigma = (w <= w_p) ? 0.07f : 0.09f;
    const float r = expf(-(w - w_p) * (w - w_p) * fast_rcp(2.0f * sigma * sigma * w_p * w_p));
    const float w2 = w * w, q = w_p * rw, q2 = q * q;
    const float jonswap = (pc.alpha * kG * kG) * fast_rcp(w2 * w2 * w) * expf(-1.25f * (q2 * q2)) * fast_exp2(r * 1.7224660244710912f);  // 3.3^r
    const float w_h = fminf(w * sqrtf(pc.depth * (1.0f / kG)), 2.0f);
    const float kit = (w_h <= 1.0f) ? 0.5f * w_h * w_h : 1.0f - 0.5f * (2.0f - w_h) * (2.0f - w_h);
    const float s = jonswap * kit;
    // hasselmann_directional_spread (:81-86) + longuet_higgins (:69-78)
    const float pr = fabsf(w * fast_rcp(w_p));
    float sh = (w <= w_p) ? 6.97f * pow_pos(pr, 4.06f) : 9.77f * pow_pos(pr, -2.33f - 1.45f * (pc.wind_speed * w_p * (1.0f / kG) - 1.17f));
    sh = sh + 16.0f * tanhf(q) * pc.swell * pc.swell;
    const float sq = sqrtf(sh);
    const float lh_norm = (sh < 0.4f) ? (0.5f / kPi) + sh * (0.220636f + sh * (-0.109f + sh * 0.090f))
                                      : (1.0f / sqrtf(kPi)) * (sq * 0.5f + fast_rcp(sq) * 0.0625f);
    // (the general cosf here: where theta - angle passes pi the cosine goes through zero and |cos|^(2 sh) with a small exponent turns its RELATIVE error
    //  into the result's -- 8 % at the texel straight downwind of a -270 degree wind with the 1.3e-7-absolute sincos_phase; tests/test_emul.py)
    const float hd = lh_norm * pow_pos(fabsf(cosf((theta - pc.angle) * 0.5f)), 2.0f * sh);
    const float am = 1.0f - pc.spread;
    const float d = ((0.5f / kPi) * (1.0f - am) + hd * am) * expf(-(1.0f - pc.detail) * (1.0f - pc.detail) * k * k);
    // gaussian(hash(id + seed)) (:44-49)
    float u1, u2;
    hash_uniform((uint32_t)(idx + pc.seed_x), (uint32_t)(idy + pc.seed_y), u1, u2);  // (its division by 2^31 is exact either way)
    const float rr = sqrtf(-2.0f * logf(u1)), th = (2.0f * kPi) * u2;
    float sn, cs;
    sincos_phase(th, sn, cs);
    const float amp = sqrtf(2.0f * s * d * w_norm);
    return cplx{rr * cs * amp, rr * sn * amp};
}

// The reference texel (spectrum_compute.glsl:117-125) is (amplitude(id), conj(amplitude(mod(-id, dims)))); only
// the first half is stored (plane a[y][x]), the second half IS the first half of the mirrored texel (Pass1).
OW_DEV f32x4 spectrum_texel(int x, int y, int n, const SpectrumPC &pc) {
    const cplx a = spectrum_amplitude(x, y, n, pc);
    const cplx b = spectrum_amplitude((n - x) % n, (n - y) % n, n, pc);
    return f32x4{a.x, a.y, b.x, -b.y};
}

}  // namespace ow
