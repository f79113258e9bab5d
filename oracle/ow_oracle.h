/*
 * ow_oracle.h -- CPU ORACLE for the ocean-wave hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C FP32 restatement of the six compute shaders and the host
 * math of 2Retr0/GodotOceanWaves (reference @ 2025-05-09).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it; the
 * product path (godotoceanwaves_amd/libocean_waves.so) never links or loads it.
 *
 * PARITY PINNING: the reference ships no tests, golden vectors or fixtures and
 * its GLSL cannot execute in this container (no Godot/Vulkan/glslang), so the
 * oracle is pinned two ways instead (see DESIGN.md "Oracle"):
 *   1. oracle/_ref/libglsl_ref.so -- the reference's own .glsl sources compiled
 *      as C++ through oracle/glsl_shim.h (built by oracle/Makefile from the
 *      files where they lie under /root/reference) and run stage by stage
 *      against this restatement; fixtures committed under tests/golden/.
 *   2. an independent NumPy FP64 twin (tests/np_twin.py) + FFT identities.
 * STATUS: pinned -- tests/test_oracle_ref.py holds every stage (spectrum texture,
 * butterfly table, fft_buffer, RGBA16F maps incl. the FP16 foam recurrence)
 * BIT-exact against (1) and against the committed golden vectors.
 *
 * Arithmetic contract (this is what "the reference result" means here):
 *   - IEEE binary32 +,-,*,/,sqrt, no FMA contraction (-ffp-contract=off),
 *     operation order exactly as written in the shaders.
 *   - libm logf/expf/powf/sinf/cosf/atan2f/tanhf for the one-time spectrum.
 *   - the per-frame dispersion relation uses the correctly-rounded tanh
 *     (float)tanh((double)a): omega then is bit-reproducible on any IEEE
 *     machine, which matters because phase = omega*t amplifies a 1-ulp
 *     difference in omega by ~1e3..1e4 rad.
 *   - RGBA16F stores round to nearest even; foam is re-read from FP16.
 */
#ifndef OW_ORACLE_H
#define OW_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OWO_NUM_SPECTRA 4

/* push-constant block of spectrum_compute.glsl:18-30 */
typedef struct {
    int32_t seed[2];
    float tile_length[2];
    float alpha;
    float peak_frequency;
    float wind_speed;
    float angle; /* radians */
    float depth;
    float swell;
    float detail;
    float spread;
} owo_spectrum_pc;

/* wave_generator.gd:116-121 (FP64 host math) */
double owo_jonswap_alpha(double wind_speed, double fetch_length_m);
double owo_jonswap_peak_angular_frequency(double wind_speed, double fetch_length_m);

/* spectrum_compute.glsl:34-41 */
void owo_hash(uint32_t x, uint32_t y, float out[2]);
/* spectrum_compute.glsl:44-49 */
void owo_gaussian(const float u[2], float out[2]);

/* IEEE half conversions used for RGBA16F image stores / loads */
uint16_t owo_f32_to_f16(float f);
float owo_f16_to_f32(uint16_t h);

/* spectrum_compute.glsl:117-125 : spectrum[y][x] = float4 (h0(id), conj(h0(-id))) */
void owo_spectrum_compute(int n, const owo_spectrum_pc *pc, float *spectrum /* n*n*4 */);

/* spectrum_modulate.glsl:48-50,60-65 : omega[y][x] (FP32) as the modulate stage sees it */
void owo_omega(int n, float tile_x, float tile_y, float depth, float *omega /* n*n */);

/* spectrum_modulate.glsl:53-90 : fft[layer][y][x] complex (interleaved re,im) */
void owo_spectrum_modulate(int n, float tile_x, float tile_y, float depth, float time,
                           const float *spectrum, float *fft /* 4*n*n*2 */);

/* fft_butterfly.glsl:19-35 : table[stage][col] = (r0 bits, r1 bits, tw.re, tw.im) */
void owo_fft_butterfly(int n, float *table /* log2n*n*4 */);

/* fft_compute.glsl:37-60 : row-wise unnormalised inverse DFT of 4 layers */
void owo_fft_rows(int n, const float *table, const float *in, float *out /* 4*n*n*2 */);

/* transpose.glsl:29-41 */
void owo_transpose(int n, const float *in, float *out /* 4*n*n*2 */);

/* wave_generator.gd:77-82 : rows -> transpose -> rows (no second transpose) */
void owo_ifft2(int n, const float *table, float *fft_half0, float *fft_half1);

/* fft_unpack.glsl:33-70.  normal is read (foam, .a) and rewritten.
 * f32_out (optional, may be NULL): 8 pre-quantisation channels per texel
 *   [hx, hy, hz, grad_x, grad_y, dhx_dx, foam, jacobian]. */
void owo_unpack(int n, const float *fft_half1, float whitecap, float foam_grow_rate,
                float foam_decay_rate, uint16_t *displacement /* n*n*4 */,
                uint16_t *normal /* n*n*4 */, float *f32_out /* n*n*8 or NULL */);

/* ---- WaveGenerator restatement (wave_generator.gd:17-109) -------------------------- */

/* wave_cascade_parameters.gd:7-42.  GDScript `float` members are FP64 and stay FP64 until render_context.gd:131-134 narrows them
 * into the push constant (after the host math of wave_generator.gd:69-71,104-106); tile_length is a Vector2 (FP32 components). */
typedef struct {
    float tile_length[2];
    double displacement_scale, normal_scale;
    double wind_speed, wind_direction /* deg */, fetch_length /* km */;
    double swell, spread, detail, whitecap, foam_amount;
    int32_t spectrum_seed[2];
    int32_t should_generate_spectrum;
    double time;
    double foam_grow_rate, foam_decay_rate;
} owo_cascade_params;

typedef struct owo_generator owo_generator;

owo_generator *owo_generator_create(int map_size, int num_cascades, float depth);
void owo_generator_destroy(owo_generator *g);
/* wave_generator.gd:65-85 for one cascade (params are mutated like the reference does) */
void owo_generator_update_cascade(owo_generator *g, int cascade, owo_cascade_params *p);
/* wave_generator.gd:101-106 */
void owo_generator_advance(owo_cascade_params *p, int count, double delta);
const float *owo_generator_spectrum(const owo_generator *g, int cascade);
const float *owo_generator_fft_half1(const owo_generator *g, int cascade);
const uint16_t *owo_generator_displacement(const owo_generator *g, int cascade);
const uint16_t *owo_generator_normal(const owo_generator *g, int cascade);
const float *owo_generator_f32(const owo_generator *g, int cascade);
/* inject foam state (FP16 bits of normal.a), used by state save/restore tests */
void owo_generator_set_normal(owo_generator *g, int cascade, const uint16_t *normal);
int owo_num_threads(void);

/* consumer side: water.gdshader:27-39,72-82 (bilinear branch) and sea_spray_particle.gdshader:78-96 at one point */
typedef struct {
    float displacement[3];
    float gradient[2];
    float gradient_scaled[2];
    float foam;
    float normal_factor, foam_factor, scale_factor;
    int32_t spray_active;
    float gradient_fragment[2]; /* water.gdshader:81, bicubic / bilinear mix included */
    float foam_fragment;
    float reserved;
} owo_surface_sample;
/* displacements / normals: [num_cascades][n][n][4] FP16 bits; map_scales: 4 floats per cascade (water.gd:105-109) */
void owo_sample_surface(int n, int num_cascades, const uint16_t *displacements, const uint16_t *normals,
                        const float *map_scales, const float *world_xz, int count, owo_surface_sample *out);

#ifdef __cplusplus
}
#endif
#endif
