/*
 * ow_oracle.c -- CPU ORACLE (test infrastructure only; see ow_oracle.h).
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the reference checkout of 2Retr0/GodotOceanWaves).  Build with
 *   gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp   (oracle/Makefile)
 * so that no multiply-add is fused and the operation order below is the
 * operation order executed.
 */
#include "ow_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* GLSL float literals: `#define PI (3.141592653589793)` and `#define G (9.81)` are
 * 32-bit floats in GLSL (spectrum_compute.glsl:11-12, spectrum_modulate.glsl:12-13). */
static const float PI_F = 3.141592653589793f;
static const float G_F = 9.81f;

int owo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- host math: assets/water/wave_generator.gd:116-121 (GDScript float = FP64) ---- */
double owo_jonswap_alpha(double wind_speed, double fetch_length_m) {
    const double g = 9.81;
    return 0.076 * pow(pow(wind_speed, 2.0) / (fetch_length_m * g), 0.22); /* `wind_speed**2`: GDScript's ** on floats is pow() */
}

double owo_jonswap_peak_angular_frequency(double wind_speed, double fetch_length_m) {
    const double g = 9.81;
    return 22.0 * pow(g * g / (wind_speed * fetch_length_m), 1.0 / 3.0);
}

/* ---- IEEE binary16 <-> binary32, round-to-nearest-even ------------------------------ */
uint16_t owo_f32_to_f16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t mag = x & 0x7FFFFFFFu;
    if (mag >= 0x7F800000u) { /* inf / nan */
        return (uint16_t)(sign | 0x7C00u | ((mag > 0x7F800000u) ? 0x0200u : 0u));
    }
    if (mag >= 0x477FF000u) { /* rounds to >= 65520 -> inf */
        return (uint16_t)(sign | 0x7C00u);
    }
    if (mag < 0x38800000u) { /* subnormal half or zero: |f| < 2^-14 */
        if (mag < 0x33000000u) return (uint16_t)sign; /* < 2^-25 -> 0 */
        int e = (int)(mag >> 23);                      /* biased exponent, 102..112 */
        uint32_t m = (mag & 0x7FFFFFu) | 0x800000u;    /* 24-bit significand */
        int shift = 126 - e;                           /* 14..24 : value = m * 2^(e-150); half sub ulp = 2^-24 */
        uint32_t q = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (q & 1u))) q++;
        return (uint16_t)(sign | q);
    }
    /* normal half */
    uint32_t e = (mag >> 23) - 112u; /* rebias 127 -> 15 */
    uint32_t m = mag & 0x7FFFFFu;
    uint32_t q = (e << 10) | (m >> 13);
    uint32_t rem = m & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (q & 1u))) q++;
    return (uint16_t)(sign | q);
}

float owo_f16_to_f32(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1Fu;
    uint32_t m = h & 0x3FFu;
    uint32_t x;
    if (e == 0) {
        if (m == 0) {
            x = sign;
        } else {
            float v = (float)m * 5.9604644775390625e-08f; /* m * 2^-24, exact */
            memcpy(&x, &v, 4);
            x |= sign;
        }
    } else if (e == 31) {
        x = sign | 0x7F800000u | (m << 13);
    } else {
        x = sign | ((e + 112u) << 23) | (m << 13);
    }
    float f;
    memcpy(&f, &x, 4);
    return f;
}

/* ---- spectrum_compute.glsl ---------------------------------------------------------- */

/* spectrum_compute.glsl:34-41 */
void owo_hash(uint32_t x, uint32_t y, float out[2]) {
    uint32_t h32 = y + 374761393u + x * 3266489917u;
    h32 = 2246822519u * (h32 ^ (h32 >> 15));
    h32 = 3266489917u * (h32 ^ (h32 >> 13));
    uint32_t n = h32 ^ (h32 >> 16);
    uint32_t rz0 = n, rz1 = n * 48271u;
    const float denom = (float)0x7FFFFFFF; /* float(0x7FFFFFFF) == 2147483648.0f */
    out[0] = (float)((rz0 >> 1) & 0x7FFFFFFFu) / denom;
    out[1] = (float)((rz1 >> 1) & 0x7FFFFFFFu) / denom;
}

/* spectrum_compute.glsl:44-49 */
void owo_gaussian(const float u[2], float out[2]) {
    float r = sqrtf(-2.0f * logf(u[0]));
    float theta = (2.0f * PI_F) * u[1];
    out[0] = r * cosf(theta);
    out[1] = r * sinf(theta);
}

/* spectrum_compute.glsl:58-66 */
static void dispersion_relation2(float k, float depth, float *w, float *dw) {
    float a = k * depth;
    float b = tanhf(a);
    float disp = sqrtf(G_F * k * b);
    float d_disp = (0.5f * G_F) * (b + a * (1.0f - b * b)) / disp;
    *w = disp;
    *dw = d_disp;
}

/* spectrum_compute.glsl:69-73 */
static float longuet_higgins_normalization(float s) {
    float a = sqrtf(s);
    if (s < 0.4f) {
        return (0.5f / PI_F) + s * (0.220636f + s * (-0.109f + s * 0.090f));
    }
    return (1.0f / sqrtf(PI_F)) * (a * 0.5f + (1.0f / a) * 0.0625f);
}

/* spectrum_compute.glsl:76-78 */
static float longuet_higgins_function(float s, float theta) {
    return longuet_higgins_normalization(s) * powf(fabsf(cosf(theta * 0.5f)), 2.0f * s);
}

/* spectrum_compute.glsl:81-86 */
static float hasselmann_directional_spread(const owo_spectrum_pc *pc, float w, float w_p, float theta) {
    float p = w / w_p;
    float s = (w <= w_p)
                  ? 6.97f * powf(fabsf(p), 4.06f)
                  : 9.77f * powf(fabsf(p), -2.33f - 1.45f * (pc->wind_speed * w_p / G_F - 1.17f));
    float s_xi = 16.0f * tanhf(w_p / w) * pc->swell * pc->swell;
    return longuet_higgins_function(s + s_xi, theta - pc->angle);
}

/* spectrum_compute.glsl:89-101 */
static float tma_spectrum(const owo_spectrum_pc *pc, float w, float w_p, float alpha) {
    const float beta = 1.25f;
    const float gamma = 3.3f;
    float sigma = (w <= w_p) ? 0.07f : 0.09f;
    float r = expf(-(w - w_p) * (w - w_p) / (2.0f * sigma * sigma * w_p * w_p));
    float jonswap = (alpha * G_F * G_F) / powf(w, 5.0f) * expf(-beta * powf(w_p / w, 4.0f)) * powf(gamma, r);
    float w_h = fminf(w * sqrtf(pc->depth / G_F), 2.0f);
    float kit = (w_h <= 1.0f) ? 0.5f * w_h * w_h : 1.0f - 0.5f * (2.0f - w_h) * (2.0f - w_h);
    return jonswap * kit;
}

/* spectrum_compute.glsl:103-115 */
static void get_spectrum_amplitude(const owo_spectrum_pc *pc, int idx, int idy, int n, float out[2]) {
    float dkx = (2.0f * PI_F) / pc->tile_length[0];
    float dky = (2.0f * PI_F) / pc->tile_length[1];
    float half = (float)n * 0.5f;
    float kx = ((float)idx - half) * dkx;
    float ky = ((float)idy - half) * dky;
    float k = sqrtf(kx * kx + ky * ky) + 1e-6f;
    float theta = atan2f(kx, ky); /* GLSL atan(y=k_vec.x, x=k_vec.y) */

    float w, dw;
    dispersion_relation2(k, pc->depth, &w, &dw);
    float w_norm = dw / k * dkx * dky;
    float s = tma_spectrum(pc, w, pc->peak_frequency, pc->alpha);
    float hd = hasselmann_directional_spread(pc, w, pc->peak_frequency, theta);
    float a = 1.0f - pc->spread;
    float mixv = (0.5f / PI_F) * (1.0f - a) + hd * a; /* GLSL mix(x,y,a) = x*(1-a)+y*a */
    float d = mixv * expf(-(1.0f - pc->detail) * (1.0f - pc->detail) * k * k);

    float u[2], g[2];
    owo_hash((uint32_t)(idx + pc->seed[0]), (uint32_t)(idy + pc->seed[1]), u);
    owo_gaussian(u, g);
    float amp = sqrtf(2.0f * s * d * w_norm);
    out[0] = g[0] * amp;
    out[1] = g[1] * amp;
}

/* spectrum_compute.glsl:117-125 */
void owo_spectrum_compute(int n, const owo_spectrum_pc *pc, float *spectrum) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < n; ++y) {
        for (int x = 0; x < n; ++x) {
            int x1 = (n - x) % n, y1 = (n - y) % n; /* ivec2(mod(-id0, dims)) */
            float a[2], b[2];
            get_spectrum_amplitude(pc, x, y, n, a);
            get_spectrum_amplitude(pc, x1, y1, n, b);
            float *o = spectrum + ((size_t)y * n + x) * 4;
            o[0] = a[0];
            o[1] = a[1];
            o[2] = b[0];
            o[3] = -b[1];
        }
    }
}

/* ---- spectrum_modulate.glsl --------------------------------------------------------- */

/* spectrum_modulate.glsl:60-61 : k_vec = (id.xy - dims*0.5)*2.0*PI / tile_length */
static inline float modulate_kcomp(int id, int n, float tile) {
    return ((((float)id - (float)n * 0.5f) * 2.0f) * PI_F) / tile;
}

/* spectrum_modulate.glsl:48-50 with the correctly-rounded tanh (see header contract) */
static inline float modulate_omega(float k, float depth) {
    float a = k * depth;
    float b = (float)tanh((double)a);
    return sqrtf(G_F * k * b);
}

void owo_omega(int n, float tile_x, float tile_y, float depth, float *omega) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < n; ++y) {
        float ky = modulate_kcomp(y, n, tile_y);
        for (int x = 0; x < n; ++x) {
            float kx = modulate_kcomp(x, n, tile_x);
            float k = sqrtf(kx * kx + ky * ky) + 1e-6f;
            omega[(size_t)y * n + x] = modulate_omega(k, depth);
        }
    }
}

/* spectrum_modulate.glsl:53-90 */
void owo_spectrum_modulate(int n, float tile_x, float tile_y, float depth, float time,
                           const float *spectrum, float *fft) {
    const size_t plane = (size_t)n * n;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < n; ++y) {
        float ky = modulate_kcomp(y, n, tile_y);
        for (int x = 0; x < n; ++x) {
            float kx = modulate_kcomp(x, n, tile_x);
            float k = sqrtf(kx * kx + ky * ky) + 1e-6f;
            float ux = kx / k, uy = ky / k;

            const float *h0 = spectrum + ((size_t)y * n + x) * 4;
            float dispersion = modulate_omega(k, depth) * time;
            float mc = cosf(dispersion), ms = sinf(dispersion);
            /* h = mul_complex(h0.xy, m) + mul_complex(h0.zw, conj(m)) */
            float a_re = h0[0] * mc - h0[1] * ms, a_im = h0[0] * ms + h0[1] * mc;
            float b_re = h0[2] * mc - h0[3] * (-ms), b_im = h0[2] * (-ms) + h0[3] * mc;
            float h_re = a_re + b_re, h_im = a_im + b_im;
            float hi_re = -h_im, hi_im = h_re; /* h_inv */

            float hx_re = hi_re * uy, hx_im = hi_im * uy;
            float hy_re = h_re, hy_im = h_im;
            float hz_re = hi_re * ux, hz_im = hi_im * ux;

            float dhy_dx_re = hi_re * ky, dhy_dx_im = hi_im * ky;
            float dhy_dz_re = hi_re * kx, dhy_dz_im = hi_im * kx;
            float dhx_dx_re = -h_re * ky * uy, dhx_dx_im = -h_im * ky * uy;
            float dhz_dz_re = -h_re * kx * ux, dhz_dz_im = -h_im * kx * ux;
            float dhz_dx_re = -h_re * ky * ux, dhz_dx_im = -h_im * ky * ux;

            size_t o = ((size_t)y * n + x) * 2;
            float *l0 = fft + 0 * plane * 2 + o, *l1 = fft + 1 * plane * 2 + o;
            float *l2 = fft + 2 * plane * 2 + o, *l3 = fft + 3 * plane * 2 + o;
            l0[0] = hx_re - hy_im;         l0[1] = hx_im + hy_re;
            l1[0] = hz_re - dhy_dx_im;     l1[1] = hz_im + dhy_dx_re;
            l2[0] = dhy_dz_re - dhx_dx_im; l2[1] = dhy_dz_im + dhx_dx_re;
            l3[0] = dhz_dz_re - dhz_dx_im; l3[1] = dhz_dz_im + dhz_dx_re;
        }
    }
}

/* ---- fft_butterfly.glsl:19-35 -------------------------------------------------------- */
static int ilog2(int n) {
    int s = 0;
    while ((1 << s) < n) ++s;
    return s;
}

void owo_fft_butterfly(int n, float *table) {
    int stages = ilog2(n);
    for (int stage = 0; stage < stages; ++stage) {
        for (int col = 0; col < n / 2; ++col) {
            uint32_t stride = 1u << stage, mid = (uint32_t)n >> (stage + 1);
            uint32_t i = (uint32_t)col >> stage, j = (uint32_t)col % stride;
            float ang = PI_F / (float)stride * (float)j;
            float twr = cosf(ang), twi = sinf(ang);
            uint32_t r0 = stride * (i + 0) + j, r1 = stride * (i + mid) + j;
            uint32_t w0 = stride * (2 * i + 0) + j, w1 = stride * (2 * i + 1) + j;
            float *e0 = table + ((size_t)stage * n + w0) * 4;
            float *e1 = table + ((size_t)stage * n + w1) * 4;
            memcpy(e0 + 0, &r0, 4); memcpy(e0 + 1, &r1, 4); /* uintBitsToFloat */
            e0[2] = twr; e0[3] = twi;
            memcpy(e1 + 0, &r0, 4); memcpy(e1 + 1, &r1, 4);
            e1[2] = -twr; e1[3] = -twi;
        }
    }
}

/* ---- fft_compute.glsl:37-60 ----------------------------------------------------------- */
void owo_fft_rows(int n, const float *table, const float *in, float *out) {
    int stages = ilog2(n);
    const size_t plane = (size_t)n * n;
#pragma omp parallel
    {
        float *buf = (float *)malloc(sizeof(float) * 2 * 2 * (size_t)n); /* ping-pong row_shared */
#pragma omp for schedule(static) collapse(2)
        for (int layer = 0; layer < OWO_NUM_SPECTRA; ++layer) {
            for (int row = 0; row < n; ++row) {
                const float *src = in + ((size_t)layer * plane + (size_t)row * n) * 2;
                float *dst = out + ((size_t)layer * plane + (size_t)row * n) * 2;
                float *pp[2] = {buf, buf + 2 * (size_t)n};
                memcpy(pp[0], src, sizeof(float) * 2 * (size_t)n);
                for (int stage = 0; stage < stages; ++stage) {
                    const float *rd = pp[stage % 2];
                    float *wr = pp[(stage + 1) % 2];
                    const float *bt = table + (size_t)stage * n * 4;
                    for (int col = 0; col < n; ++col) {
                        uint32_t r0, r1;
                        memcpy(&r0, bt + (size_t)col * 4 + 0, 4); /* floatBitsToUint */
                        memcpy(&r1, bt + (size_t)col * 4 + 1, 4);
                        float twr = bt[(size_t)col * 4 + 2], twi = bt[(size_t)col * 4 + 3];
                        float ur = rd[2 * r0], ui = rd[2 * r0 + 1];
                        float lr = rd[2 * r1], li = rd[2 * r1 + 1];
                        /* upper + mul_complex(lower, twiddle) */
                        wr[2 * col] = ur + (lr * twr - li * twi);
                        wr[2 * col + 1] = ui + (lr * twi + li * twr);
                    }
                }
                memcpy(dst, pp[stages % 2], sizeof(float) * 2 * (size_t)n);
            }
        }
        free(buf);
    }
}

/* ---- transpose.glsl:29-41 --------------------------------------------------------------- */
void owo_transpose(int n, const float *in, float *out) {
    const size_t plane = (size_t)n * n;
#pragma omp parallel for schedule(static) collapse(2)
    for (int layer = 0; layer < OWO_NUM_SPECTRA; ++layer) {
        for (int by = 0; by < n; by += 32) {
            for (int bx = 0; bx < n; bx += 32) {
                for (int y = by; y < by + 32 && y < n; ++y) {
                    for (int x = bx; x < bx + 32 && x < n; ++x) {
                        const float *s = in + ((size_t)layer * plane + (size_t)y * n + x) * 2;
                        float *d = out + ((size_t)layer * plane + (size_t)x * n + y) * 2;
                        d[0] = s[0];
                        d[1] = s[1];
                    }
                }
            }
        }
    }
}

/* ---- wave_generator.gd:77-82 ------------------------------------------------------------ */
void owo_ifft2(int n, const float *table, float *half0, float *half1) {
    owo_fft_rows(n, table, half0, half1); /* fft_compute : half0 -> half1 */
    owo_transpose(n, half1, half0);       /* transpose   : half1 -> half0 */
    owo_fft_rows(n, table, half0, half1); /* fft_compute : half0 -> half1 */
}

/* ---- fft_unpack.glsl:33-70 ---------------------------------------------------------------- */
void owo_unpack(int n, const float *fft, float whitecap, float foam_grow_rate, float foam_decay_rate,
                uint16_t *displacement, uint16_t *normal, float *f32_out) {
    const size_t plane = (size_t)n * n;
    const float decay = expf(-foam_decay_rate);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < n; ++y) {
        for (int x = 0; x < n; ++x) {
            size_t t = (size_t)y * n + x;
            float sign_shift = (float)(-2 * ((x & 1) ^ (y & 1)) + 1);
            const float *l0 = fft + (0 * plane + t) * 2, *l1 = fft + (1 * plane + t) * 2;
            const float *l2 = fft + (2 * plane + t) * 2, *l3 = fft + (3 * plane + t) * 2;

            float hx = l0[0] * sign_shift, hy = l0[1] * sign_shift, hz = l1[0] * sign_shift;
            displacement[t * 4 + 0] = owo_f32_to_f16(hx);
            displacement[t * 4 + 1] = owo_f32_to_f16(hy);
            displacement[t * 4 + 2] = owo_f32_to_f16(hz);
            displacement[t * 4 + 3] = owo_f32_to_f16(0.0f * sign_shift);

            float dhy_dx = l1[1] * sign_shift;
            float dhy_dz = l2[0] * sign_shift;
            float dhx_dx = l2[1] * sign_shift;
            float dhz_dz = l3[0] * sign_shift;
            float dhz_dx = l3[1] * sign_shift;

            float jacobian = (1.0f + dhx_dx) * (1.0f + dhz_dz) - dhz_dx * dhz_dx;
            float foam_factor = -fminf(0.0f, jacobian - whitecap);
            float foam = owo_f16_to_f32(normal[t * 4 + 3]);
            foam *= decay;
            foam += foam_factor * foam_grow_rate;
            foam = fminf(fmaxf(foam, 0.0f), 1.0f);

            float gx = dhy_dx / (1.0f + fabsf(dhx_dx));
            float gy = dhy_dz / (1.0f + fabsf(dhz_dz));
            normal[t * 4 + 0] = owo_f32_to_f16(gx);
            normal[t * 4 + 1] = owo_f32_to_f16(gy);
            normal[t * 4 + 2] = owo_f32_to_f16(dhx_dx);
            normal[t * 4 + 3] = owo_f32_to_f16(foam);
            if (f32_out) {
                float *o = f32_out + t * 8;
                o[0] = hx; o[1] = hy; o[2] = hz; o[3] = gx; o[4] = gy; o[5] = dhx_dx; o[6] = foam; o[7] = jacobian;
            }
        }
    }
}

/* ---- WaveGenerator (wave_generator.gd:17-109) ------------------------------------------------ */
struct owo_generator {
    int n, cascades;
    float depth;
    float *table;     /* butterfly_factors */
    float *spectrum;  /* cascades * n*n*4 */
    float *fft;       /* cascades * 2 halves * 4*n*n*2 */
    uint16_t *disp;   /* cascades * n*n*4 */
    uint16_t *normal; /* cascades * n*n*4 */
    float *f32;       /* cascades * n*n*8 */
};

owo_generator *owo_generator_create(int map_size, int num_cascades, float depth) {
    owo_generator *g = (owo_generator *)calloc(1, sizeof(*g));
    size_t nn = (size_t)map_size * map_size;
    g->n = map_size;
    g->cascades = num_cascades;
    g->depth = depth;
    g->table = (float *)calloc((size_t)ilog2(map_size) * map_size * 4, sizeof(float));
    g->spectrum = (float *)calloc(nn * 4 * num_cascades, sizeof(float));
    g->fft = (float *)calloc(nn * 4 * 2 * 2 * num_cascades, sizeof(float));
    g->disp = (uint16_t *)calloc(nn * 4 * num_cascades, sizeof(uint16_t));
    g->normal = (uint16_t *)calloc(nn * 4 * num_cascades, sizeof(uint16_t));
    g->f32 = (float *)calloc(nn * 8 * num_cascades, sizeof(float));
    owo_fft_butterfly(map_size, g->table); /* wave_generator.gd:52-54 */
    return g;
}

void owo_generator_destroy(owo_generator *g) {
    if (!g) return;
    free(g->table); free(g->spectrum); free(g->fft); free(g->disp); free(g->normal); free(g->f32);
    free(g);
}

/* wave_generator.gd:101-106 */
void owo_generator_advance(owo_cascade_params *p, int count, double delta) {
    for (int i = 0; i < count; ++i) {
        p[i].time += delta;
        p[i].foam_grow_rate = delta * p[i].foam_amount * 7.5;
        double d = 10.0 - p[i].foam_amount;
        p[i].foam_decay_rate = delta * (d > 0.5 ? d : 0.5) * 1.15;
    }
}

/* wave_generator.gd:65-85 */
void owo_generator_update_cascade(owo_generator *g, int c, owo_cascade_params *p) {
    size_t nn = (size_t)g->n * g->n;
    float *spectrum = g->spectrum + nn * 4 * c;
    float *half0 = g->fft + nn * 4 * 2 * 2 * c, *half1 = half0 + nn * 4 * 2;
    if (p->should_generate_spectrum) {
        owo_spectrum_pc pc;
        /* wave_generator.gd:69-71: FP64 host math on the FP64 parameters; render_context.gd:131-134 narrows every float to FP32
         * when packing -- and only there */
        double F = p->fetch_length * 1e3;
        pc.seed[0] = p->spectrum_seed[0]; pc.seed[1] = p->spectrum_seed[1];
        pc.tile_length[0] = p->tile_length[0]; pc.tile_length[1] = p->tile_length[1];
        pc.alpha = (float)owo_jonswap_alpha(p->wind_speed, F);
        pc.peak_frequency = (float)owo_jonswap_peak_angular_frequency(p->wind_speed, F);
        pc.wind_speed = (float)p->wind_speed;
        pc.angle = (float)(p->wind_direction * (3.14159265358979323846 / 180.0)); /* deg_to_rad */
        pc.depth = g->depth;
        pc.swell = (float)p->swell; pc.detail = (float)p->detail; pc.spread = (float)p->spread;
        owo_spectrum_compute(g->n, &pc, spectrum);
        p->should_generate_spectrum = 0;
    }
    owo_spectrum_modulate(g->n, p->tile_length[0], p->tile_length[1], g->depth, (float)p->time, spectrum, half0);
    owo_ifft2(g->n, g->table, half0, half1);
    owo_unpack(g->n, half1, (float)p->whitecap, (float)p->foam_grow_rate, (float)p->foam_decay_rate,
               g->disp + nn * 4 * c, g->normal + nn * 4 * c, g->f32 + nn * 8 * c);
}

const float *owo_generator_spectrum(const owo_generator *g, int c) { return g->spectrum + (size_t)g->n * g->n * 4 * c; }
const float *owo_generator_fft_half1(const owo_generator *g, int c) {
    return g->fft + (size_t)g->n * g->n * 4 * 2 * 2 * c + (size_t)g->n * g->n * 4 * 2;
}
const uint16_t *owo_generator_displacement(const owo_generator *g, int c) { return g->disp + (size_t)g->n * g->n * 4 * c; }
const uint16_t *owo_generator_normal(const owo_generator *g, int c) { return g->normal + (size_t)g->n * g->n * 4 * c; }
const float *owo_generator_f32(const owo_generator *g, int c) { return g->f32 + (size_t)g->n * g->n * 8 * c; }
void owo_generator_set_normal(owo_generator *g, int c, const uint16_t *normal) {
    memcpy(g->normal + (size_t)g->n * g->n * 4 * c, normal, (size_t)g->n * g->n * 4 * sizeof(uint16_t));
}

/* ---- consumer side: what the spatial and particle shaders read back --------------- */

/* texture(sampler2DArray, vec3(uv, layer)) with GL_LINEAR + GL_REPEAT on an n x n RGBA16F layer (OpenGL 4.6 core
 * spec 8.14.2 / Vulkan spec 16.8: unnormalised coordinate u*n - 0.5, i0 = floor, weights = fract, exact FP32
 * weights -- real texture units quantise them to 8 bits, which the reference does not pin) */
static void texture_linear_repeat(const uint16_t *layer, int n, float u, float v, float out[4]) {
    float un = u * (float)n - 0.5f, vn = v * (float)n - 0.5f;
    float fi = floorf(un), fj = floorf(vn);
    float a = un - fi, b = vn - fj;
    long i0 = (long)fi % n, j0 = (long)fj % n;
    if (i0 < 0) i0 += n;
    if (j0 < 0) j0 += n;
    long i1 = (i0 + 1) % n, j1 = (j0 + 1) % n;
    for (int k = 0; k < 4; ++k) {
        float t00 = owo_f16_to_f32(layer[((size_t)j0 * n + i0) * 4 + k]), t10 = owo_f16_to_f32(layer[((size_t)j0 * n + i1) * 4 + k]);
        float t01 = owo_f16_to_f32(layer[((size_t)j1 * n + i0) * 4 + k]), t11 = owo_f16_to_f32(layer[((size_t)j1 * n + i1) * 4 + k]);
        out[k] = (t00 * (1.0f - a) + t10 * a) * (1.0f - b) + (t01 * (1.0f - a) + t11 * a) * b;
    }
}

static float mixf(float x, float y, float a) { return x * (1.0f - a) + y * a; } /* GLSL mix() */

/* water.gdshader:41-51 cubic_weights, :53-68 texture_bicubic: cubic B-spline filtering as four bilinear taps */
static void cubic_weights(float a, float w[4]) {
    float a2 = a * a, a3 = a2 * a;
    w[0] = (-a3 + a2 * 3.0f - a * 3.0f + 1.0f) / 6.0f;
    w[1] = (a3 * 3.0f - a2 * 6.0f + 4.0f) / 6.0f;
    w[2] = (-a3 * 3.0f + a2 * 3.0f + a * 3.0f + 1.0f) / 6.0f;
    w[3] = a3 / 6.0f;
}
static void texture_bicubic(const uint16_t *layer, int n, float u, float v, float out[4]) {
    const float dims = (float)n, dims_inv = 1.0f / dims;
    const float x = u * dims + 0.5f, y = v * dims + 0.5f;
    const float fx = x - floorf(x), fy = y - floorf(y);
    float wx[4], wy[4];
    cubic_weights(fx, wx);
    cubic_weights(fy, wy);
    const float gx0 = wx[0] + wx[1], gx1 = wx[2] + wx[3], gy0 = wy[0] + wy[1], gy1 = wy[2] + wy[3];
    const float hx0 = (wx[1] / gx0 + -1.5f + floorf(x)) * dims_inv, hx1 = (wx[3] / gx1 + 0.5f + floorf(x)) * dims_inv;
    const float hy0 = (wy[1] / gy0 + -1.5f + floorf(y)) * dims_inv, hy1 = (wy[3] / gy1 + 0.5f + floorf(y)) * dims_inv;
    const float wgx = gx0 / (gx0 + gx1), wgy = gy0 / (gy0 + gy1);
    float t_yw[4], t_xw[4], t_yz[4], t_xz[4]; /* texture(h.yw), (h.xw), (h.yz), (h.xz) */
    texture_linear_repeat(layer, n, hx1, hy1, t_yw);
    texture_linear_repeat(layer, n, hx0, hy1, t_xw);
    texture_linear_repeat(layer, n, hx1, hy0, t_yz);
    texture_linear_repeat(layer, n, hx0, hy0, t_xz);
    for (int k = 0; k < 4; ++k) out[k] = mixf(mixf(t_yw[k], t_xw[k], wgx), mixf(t_yz[k], t_xz[k], wgx), wgy);
}

void owo_sample_surface(int n, int num_cascades, const uint16_t *displacements, const uint16_t *normals,
                        const float *map_scales, const float *world_xz, int count, owo_surface_sample *out) {
    const size_t layer = (size_t)n * n * 4;
    for (int p = 0; p < count; ++p) {
        const float x = world_xz[2 * p], z = world_xz[2 * p + 1];
        owo_surface_sample s;
        memset(&s, 0, sizeof(s));
        for (int i = 0; i < num_cascades; ++i) {
            const float *scales = map_scales + 4 * i;
            float d[4], g[4];
            /* water.gdshader:33-36 and sea_spray_particle.gdshader:104-107 */
            texture_linear_repeat(displacements + layer * i, n, x * scales[0], z * scales[1], d);
            for (int k = 0; k < 3; ++k) s.displacement[k] += d[k] * scales[2];
            /* sea_spray_particle.gdshader:81-82 (.xyw, unscaled); water.gdshader:81 (.xyw * vec3(scales.ww, 1)) */
            texture_linear_repeat(normals + layer * i, n, x * scales[0], z * scales[1], g);
            s.gradient[0] += g[0];
            s.gradient[1] += g[1];
            s.gradient_scaled[0] += g[0] * scales[3];
            s.gradient_scaled[1] += g[1] * scales[3];
            s.foam += g[3];
            /* water.gdshader:74-82 fragment(): bicubic and bilinear mixed by the pixels per metre of this cascade */
            {
                float bc[4];
                const float ppm = (float)n * fminf(scales[0], scales[1]);
                const float a = fminf(1.0f, ppm * 0.1f);
                texture_bicubic(normals + layer * i, n, x * scales[0], z * scales[1], bc);
                s.gradient_fragment[0] += mixf(bc[0], g[0], a) * scales[3];
                s.gradient_fragment[1] += mixf(bc[1], g[1], a) * scales[3];
                s.foam_fragment += mixf(bc[3], g[3], a) * 1.0f;
            }
        }
        /* sea_spray_particle.gdshader:83-89 */
        float nx = -s.gradient[0], ny = 1.0f, nz = -s.gradient[1];
        float normal_y = ny * (1.0f / sqrtf(nx * nx + ny * ny + nz * nz));
        s.normal_factor = mixf(0.25f, 1.0f, fminf((normal_y - 0.92f) / (0.99f - 0.92f), 1.0f));
        s.foam_factor = mixf(0.25f, 1.0f, fminf((s.foam - 0.9f) / (1.0f - 0.9f), 1.0f));
        s.spray_active = (s.normal_factor >= 0.0f && s.normal_factor <= 1.0f && s.foam > 0.9f) ? 1 : 0;
        s.scale_factor = s.normal_factor * s.foam_factor;
        out[p] = s;
    }
}
