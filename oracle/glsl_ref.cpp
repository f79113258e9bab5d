// glsl_ref.cpp -- TEST INFRASTRUCTURE.  Executes the reference's own six compute shaders on the CPU.
//
// The shader sources are NOT in this repository: oracle/Makefile runs glsl_prep.py over
// /root/reference/assets/shaders/compute/*.glsl and writes C++-parsable fragments to oracle/_ref/*.inc
// (git-ignored); this file #includes those fragments, each inside its own namespace, on top of glsl_shim.h,
// and provides one C entry point per dispatch of wave_generator.gd:44-49 with the same group counts.
// Only tests/ and the golden-vector generator call it; it exists to pin oracle/ow_oracle.c to the reference.
#include <stdio.h>
#include <stdlib.h>
#include <ucontext.h>

#include <vector>

#include "glsl_shim.h"

namespace glsl {
thread_local uvec3 gl_NumWorkGroups, gl_WorkGroupID, gl_LocalInvocationID, gl_GlobalInvocationID;

// ---- workgroup execution: sequential, or cooperative fibers when the shader calls barrier() ----------------
namespace {
struct Fiber {
    ucontext_t ctx;
    bool done = false;
    uvec3 local;
};
thread_local ucontext_t g_sched;
thread_local Fiber *g_cur = nullptr;
thread_local void (*g_main)() = nullptr;
thread_local std::vector<Fiber> g_fibers;
thread_local std::vector<char> g_stacks;
constexpr size_t kStack = 64 * 1024;

void fiber_entry() {
    g_main();
    g_cur->done = true;
    swapcontext(&g_cur->ctx, &g_sched);
}
}  // namespace

void barrier() { swapcontext(&g_cur->ctx, &g_sched); }

static void set_ids(const uvec3 &wg, const uvec3 &size, const uvec3 &local) {
    gl_WorkGroupID = wg;
    gl_LocalInvocationID = local;
    gl_GlobalInvocationID = uvec3(wg.x * size.x + local.x, wg.y * size.y + local.y, wg.z * size.z + local.z);
}

// dispatch(groups) of a shader with local size `size`; uses_barrier selects the fiber scheduler
static void dispatch(void (*shader_main)(), const uvec3 &size, uint gx, uint gy, uint gz, bool uses_barrier) {
    gl_NumWorkGroups = uvec3(gx, gy, gz);
    const uint per_wg = size.x * size.y * size.z;
    if (uses_barrier) {
        if (g_fibers.size() < per_wg) g_fibers.resize(per_wg);
        if (g_stacks.size() < (size_t)per_wg * kStack) g_stacks.resize((size_t)per_wg * kStack);
    }
    for (uint wz = 0; wz < gz; ++wz)
        for (uint wy = 0; wy < gy; ++wy)
            for (uint wx = 0; wx < gx; ++wx) {
                const uvec3 wg(wx, wy, wz);
                if (!uses_barrier) {
                    for (uint lz = 0; lz < size.z; ++lz)
                        for (uint ly = 0; ly < size.y; ++ly)
                            for (uint lx = 0; lx < size.x; ++lx) {
                                set_ids(wg, size, uvec3(lx, ly, lz));
                                shader_main();
                            }
                    continue;
                }
                g_main = shader_main;
                uint i = 0;
                for (uint lz = 0; lz < size.z; ++lz)
                    for (uint ly = 0; ly < size.y; ++ly)
                        for (uint lx = 0; lx < size.x; ++lx, ++i) {
                            Fiber &f = g_fibers[i];
                            f.done = false;
                            f.local = uvec3(lx, ly, lz);
                            getcontext(&f.ctx);
                            f.ctx.uc_stack.ss_sp = g_stacks.data() + (size_t)i * kStack;
                            f.ctx.uc_stack.ss_size = kStack;
                            f.ctx.uc_link = nullptr;
                            makecontext(&f.ctx, fiber_entry, 0);
                        }
                // every live invocation runs to its next barrier() (or to its end) before any proceeds
                for (bool any = true; any;) {
                    any = false;
                    for (uint k = 0; k < per_wg; ++k) {
                        Fiber &f = g_fibers[k];
                        if (f.done) continue;
                        set_ids(wg, size, f.local);
                        g_cur = &f;
                        swapcontext(&g_sched, &f.ctx);
                        any = any || !f.done;
                    }
                }
            }
}
}  // namespace glsl

#define GLSL_USING                                                                                            \
    using glsl::vec2; using glsl::vec4; using glsl::ivec2; using glsl::ivec3; using glsl::uvec2; using glsl::uvec3; \
    using glsl::vec3; using glsl::sampler2DArray; using glsl::texture; using glsl::textureSize; using glsl::fract;  \
    using glsl::floor; using glsl::normalize;                                                                  \
    using glsl::image2DArray; using glsl::cos; using glsl::sin; using glsl::exp; using glsl::log; using glsl::sqrt; \
    using glsl::inversesqrt; using glsl::pow; using glsl::atan; using glsl::abs; using glsl::min; using glsl::max; \
    using glsl::clamp; using glsl::mix; using glsl::length; using glsl::mod; using glsl::floatBitsToUint;      \
    using glsl::uintBitsToFloat; using glsl::findMSB; using glsl::imageSize; using glsl::imageLoad;            \
    using glsl::imageStore; using glsl::barrier; using glsl::gl_NumWorkGroups; using glsl::gl_WorkGroupID;     \
    using glsl::gl_LocalInvocationID; using glsl::gl_GlobalInvocationID;

// Each shader gets its own namespace: its #defines (PI, G, TILE_SIZE, ...) are undefined again afterwards.
namespace sh_spectrum_compute {
GLSL_USING
#define tanh glsl::tanh_libm
#include "spectrum_compute.inc"
#undef tanh
#undef PI
#undef G
}  // namespace sh_spectrum_compute

namespace sh_spectrum_modulate {
GLSL_USING
#define tanh glsl::tanh_cr
#include "spectrum_modulate.inc"
#undef tanh
#undef PI
#undef G
#undef NUM_SPECTRA
#undef FFT_DATA
}  // namespace sh_spectrum_modulate

namespace sh_fft_butterfly {
GLSL_USING
#include "fft_butterfly.inc"
#undef PI
#undef BUTTERFLY
}  // namespace sh_fft_butterfly

namespace sh_fft_compute {
GLSL_USING
#include "fft_compute.inc"
#undef PI
#undef MAX_MAP_SIZE
#undef NUM_SPECTRA
#undef ROW_SHARED
#undef BUTTERFLY
#undef DATA_IN
#undef DATA_OUT
}  // namespace sh_fft_compute

namespace sh_transpose {
GLSL_USING
#include "transpose.inc"
#undef TILE_SIZE
#undef NUM_SPECTRA
#undef DATA_IN
#undef DATA_OUT
}  // namespace sh_transpose

namespace sh_fft_unpack {
GLSL_USING
#include "fft_unpack.inc"
#undef TILE_SIZE
#undef NUM_SPECTRA
#undef FFT_DATA
}  // namespace sh_fft_unpack

// ---- the map-reading parts of the spatial / particle shaders --------------------------------
// .gdshader files are not GLSL translation units; oracle/Makefile has glsl_prep.py --extract copy out, verbatim, the
// functions cubic_weights / texture_bicubic (water.gdshader:41-68), the cascade loops of vertex() (:31-37) and fragment()
// (:72-82), and of the particle shader's process() the spawn decision (sea_spray_particle.gdshader:80-89) and the
// displacement lookup (:103-107).  The uniforms and built-ins those statements name are the globals declared here.
namespace sh_water {
GLSL_USING
#include "water_gdshader.inc.defs"
vec4 map_scales[MAX_CASCADES];   // water.gdshader:18
uint num_cascades;               // :19
sampler2DArray displacements;    // :20
sampler2DArray normals;          // :21
vec2 UV;                         // built-in; vertex() sets UV = VERTEX.xz (:28) and fragment() receives it interpolated
vec3 VERTEX;                     // built-in (only its distance to the camera is taken from it in the extracted range)
vec3 out_displacement, out_gradient;
#include "water_gdshader.inc"
}  // namespace sh_water

namespace sh_spray {
GLSL_USING
vec4 map_scales[8];              // sea_spray_particle.gdshader (group_uniforms cascade_data), MAX_CASCADES entries
uint num_cascades;
sampler2DArray displacements, normals;
vec3 START_POS;                  // varying of the particle (its spawn position)
bool ACTIVE;                     // built-in
float SCALE_FACTOR;              // #define SCALE_FACTOR USERDATA1.w in the shader: a per-particle float
vec3 out_displacement, out_gradient;
float out_normal_factor, out_foam_factor;
#include "sea_spray_particle_gdshader.inc"
}  // namespace sh_spray

#undef MAX_CASCADES
#undef in
#undef shared

// ---- C entry points: one per dispatch of WaveGenerator (single cascade, cascade_index = 0) -----------------
// Buffers (all caller-owned, one cascade):
//   spectrum : n*n*4 floats (RGBA32F layer)          butterfly : log2(n)*n*4 floats
//   fft      : 2 halves * 4 layers * n*n * 2 floats   displacement / normal : n*n*4 uint16 (RGBA16F layer)
extern "C" {

struct ref_spectrum_pc {  // spectrum_compute.glsl:18-30
    int32_t seed[2];
    float tile_length[2];
    float alpha, peak_frequency, wind_speed, angle, depth, swell, detail, spread;
};

static glsl::image2DArray fp32_image(int n, float *p) {
    glsl::image2DArray im;
    im.w = im.h = n;
    im.layers = 1;
    im.fp32 = true;
    im.f = p;
    return im;
}
static glsl::image2DArray fp16_image(int n, uint16_t *p) {
    glsl::image2DArray im;
    im.w = im.h = n;
    im.layers = 1;
    im.fp32 = false;
    im.q = p;
    return im;
}

// wave_generator.gd:44,71  dispatch [N/16, N/16, 1]
void ref_spectrum_compute(int n, const ref_spectrum_pc *pc, float *spectrum) {
    namespace S = sh_spectrum_compute;
    S::spectrum = fp32_image(n, spectrum);
    S::seed = glsl::ivec2(pc->seed[0], pc->seed[1]);
    S::tile_length = glsl::vec2(pc->tile_length[0], pc->tile_length[1]);
    S::alpha = pc->alpha;
    S::peak_frequency = pc->peak_frequency;
    S::wind_speed = pc->wind_speed;
    S::angle = pc->angle;
    S::depth = pc->depth;
    S::swell = pc->swell;
    S::detail = pc->detail;
    S::spread = pc->spread;
    S::cascade_index = 0;
    glsl::dispatch(S::shader_main, S::gl_WorkGroupSize, n / 16, n / 16, 1, false);
}

// wave_generator.gd:45,73  dispatch [N/16, N/16, 1]
void ref_spectrum_modulate(int n, float tile_x, float tile_y, float depth, float time, float *spectrum, float *fft) {
    namespace S = sh_spectrum_modulate;
    S::spectrum = fp32_image(n, spectrum);
    S::data = reinterpret_cast<glsl::vec2 *>(fft);
    S::tile_length = glsl::vec2(tile_x, tile_y);
    S::depth = depth;
    S::time = time;
    S::cascade_index = 0;
    glsl::dispatch(S::shader_main, S::gl_WorkGroupSize, n / 16, n / 16, 1, false);
}

// wave_generator.gd:46,52-54  dispatch [N/128, log2 N, 1]
void ref_fft_butterfly(int n, float *butterfly) {
    namespace S = sh_fft_butterfly;
    S::butterfly = reinterpret_cast<glsl::vec4 *>(butterfly);
    int stages = 0;
    while ((1 << stages) < n) ++stages;
    glsl::dispatch(S::shader_main, S::gl_WorkGroupSize, n / 2 / 64, stages, 1, false);
}

// wave_generator.gd:47,79,82  dispatch [1, N, 4]
void ref_fft_compute(int n, float *butterfly, float *fft) {
    namespace S = sh_fft_compute;
    S::butterfly = reinterpret_cast<glsl::vec4 *>(butterfly);
    S::data = reinterpret_cast<glsl::vec2 *>(fft);
    S::cascade_index = 0;
    glsl::dispatch(S::shader_main, S::gl_WorkGroupSize, 1, n, 4, true);
}

// wave_generator.gd:48,80  dispatch [N/32, N/32, 4]
void ref_transpose(int n, float *butterfly, float *fft) {
    namespace S = sh_transpose;
    S::butterfly = reinterpret_cast<glsl::vec4 *>(butterfly);
    S::data = reinterpret_cast<glsl::vec2 *>(fft);
    S::cascade_index = 0;
    glsl::dispatch(S::shader_main, S::gl_WorkGroupSize, n / 32, n / 32, 4, true);
}

// wave_generator.gd:49,85  dispatch [N/16, N/16, 1]
void ref_fft_unpack(int n, float *fft, float whitecap, float foam_grow_rate, float foam_decay_rate, uint16_t *displacement,
                    uint16_t *normal) {
    namespace S = sh_fft_unpack;
    S::displacement_map = fp16_image(n, displacement);
    S::normal_map = fp16_image(n, normal);
    S::data = reinterpret_cast<glsl::vec2 *>(fft);
    S::cascade_index = 0;
    S::whitecap = whitecap;
    S::foam_grow_rate = foam_grow_rate;
    S::foam_decay_rate = foam_decay_rate;
    glsl::dispatch(S::shader_main, S::gl_WorkGroupSize, n / 16, n / 16, 1, true);
}

// Same record as owo_surface_sample (oracle/ow_oracle.h) / ow_surface_sample (include/ocean_waves.h).
struct ref_surface_sample {
    float displacement[3], gradient[2], gradient_scaled[2], foam, normal_factor, foam_factor, scale_factor;
    int32_t spray_active;
    float gradient_fragment[2], foam_fragment, reserved;
};

// What the reference's consumers read at world points (x, z): vertex() displacement sum, fragment() gradient / foam sum
// (bicubic / bilinear mix), the particle shader's spawn decision and its own displacement sum (must equal vertex()'s).
// gradient_scaled has no statement of its own in the reference (it is the bilinear operand of the mix in
// water.gdshader:81) and stays zero here; `displacement_particle` receives sea_spray_particle.gdshader:103-107.
void ref_sample_surface(int n, int num_cascades, const uint16_t *displacements, const uint16_t *normals, const float *map_scales,
                        const float *world_xz, int count, ref_surface_sample *out, float *displacement_particle) {
    namespace W = sh_water;
    namespace P = sh_spray;
    glsl::sampler2DArray d, m;
    d.w = d.h = m.w = m.h = n;
    d.layers = m.layers = num_cascades;
    d.q = displacements;
    m.q = normals;
    W::displacements = P::displacements = d;
    W::normals = P::normals = m;
    W::num_cascades = P::num_cascades = (uint)num_cascades;
    for (int i = 0; i < num_cascades; ++i)
        W::map_scales[i] = P::map_scales[i] = glsl::vec4(map_scales[4 * i], map_scales[4 * i + 1], map_scales[4 * i + 2], map_scales[4 * i + 3]);
    for (int p = 0; p < count; ++p) {
        const float x = world_xz[2 * p], z = world_xz[2 * p + 1];
        ref_surface_sample s;
        memset(&s, 0, sizeof(s));
        W::VERTEX = glsl::vec3(x, 0.0f, z);
        W::UV = W::VERTEX.xz;  // water.gdshader:28
        W::vertex_sum();
        W::fragment_sum();
        P::START_POS = glsl::vec3(x, 0.0f, z);
        P::spawn();
        P::follow();
        for (int k = 0; k < 3; ++k) s.displacement[k] = W::out_displacement.d[k];
        s.gradient_fragment[0] = W::out_gradient.x;
        s.gradient_fragment[1] = W::out_gradient.y;
        s.foam_fragment = W::out_gradient.z;
        s.gradient[0] = P::out_gradient.x;
        s.gradient[1] = P::out_gradient.y;
        s.foam = P::out_gradient.z;
        s.normal_factor = P::out_normal_factor;
        s.foam_factor = P::out_foam_factor;
        s.scale_factor = P::SCALE_FACTOR;
        s.spray_active = P::ACTIVE ? 1 : 0;
        out[p] = s;
        if (displacement_particle)
            for (int k = 0; k < 3; ++k) displacement_particle[3 * p + k] = P::out_displacement.d[k];
    }
}

}  // extern "C"
