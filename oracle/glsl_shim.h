// glsl_shim.h -- TEST INFRASTRUCTURE.  The subset of GLSL 4.60 that the six compute shaders of
// 2Retr0/GodotOceanWaves (assets/shaders/compute/*.glsl) and the map-reading parts of its spatial / particle shaders
// (assets/shaders/spatial/water.gdshader:27-84, sea_spray_particle.gdshader:78-107) use, as C++: vector types with the swizzles that
// occur, the built-in functions that occur, image2DArray, shared memory and barrier().  With it the
// reference's OWN shader sources (read where they lie under /root/reference, token-rewritten by glsl_prep.py
// into oracle/_ref/, never committed) compile with g++ and execute on the CPU: that build is what pins the
// hand-written oracle (oracle/ow_oracle.c) and generates tests/golden/.
//
// Semantics chosen where GLSL leaves them to the implementation:
//   * float arithmetic is IEEE binary32, no contraction (build flag -ffp-contract=off);
//   * built-ins map to glibc's float functions: cos->cosf, sin->sinf, exp->expf, log->logf, pow->powf,
//     sqrt->sqrtf, atan(y,x)->atan2f, tanh->GLSL_TANH (per shader: tanhf in spectrum_compute, the correctly
//     rounded (float)tanh((double)x) in spectrum_modulate -- the same two choices oracle/ow_oracle.c documents,
//     SURVEY.md H1), inversesqrt(x) = 1/sqrtf(x), length(v) = sqrtf(x*x + y*y), mix(x,y,a) = x*(1-a) + y*a,
//     mod(x,y) = x - y*floor(x/y)  (the GLSL specification's own formulas);
//   * rgba16f images quantise on imageStore with round-to-nearest-even; the `spectrum` image is declared
//     rgba16f in the shaders but ALLOCATED R32G32B32A32_SFLOAT by wave_generator.gd:31, so it stores FP32
//     (SURVEY.md F7);
//   * barrier(): the invocations of a workgroup run as cooperative fibers (ucontext); barrier() yields.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef unsigned int uint;

namespace glsl {

// ---- swizzle proxies ------------------------------------------------------------------------------
// (LEN = number of components of the enclosing vector: a proxy must not enlarge it -- vec2 is 8 bytes in buffers)
template <class V, class S, int LEN, int... I>
struct Swz {
    S v[LEN];
    operator V() const { return V(v[I]...); }
    Swz &operator=(const V &o) {
        const int idx[] = {I...};
        const V c(o);
        for (int k = 0; k < (int)sizeof...(I); ++k) v[idx[k]] = c.d[k];
        return *this;
    }
};
template <class V, class S, int A, int B, int LEN>
using Swz2 = Swz<V, S, LEN, A, B>;

#define GLSL_VEC_COMMON(V, S, NN)                                  \
    V(const V &o) { for (int i = 0; i < NN; ++i) d[i] = o.d[i]; } \
    V &operator=(const V &o) {                                     \
        for (int i = 0; i < NN; ++i) d[i] = o.d[i];                \
        return *this;                                              \
    }                                                              \
    S &operator[](int i) { return d[i]; }                          \
    const S &operator[](int i) const { return d[i]; }

struct vec2;
struct vec3;
struct vec4;
struct ivec2;
struct uvec2;

struct vec2 {
    union {
        struct { float x, y; };
        float d[2];
        Swz2<vec2, float, 0, 1, 2> xy;
        Swz2<vec2, float, 1, 0, 2> yx;
        Swz<vec4, float, 2, 0, 0, 1, 1> xxyy;
        Swz<vec4, float, 2, 0, 1, 0, 1> xyxy;
    };
    vec2() : x(0), y(0) {}
    vec2(float a, float b) : x(a), y(b) {}
    explicit vec2(float a) : x(a), y(a) {}
    inline explicit vec2(const uvec2 &u);
    inline vec2(const ivec2 &i);  // GLSL implicit int -> float conversion
    GLSL_VEC_COMMON(vec2, float, 2)
};
struct ivec2 {
    union {
        struct { int x, y; };
        int d[2];
        Swz2<ivec2, int, 0, 1, 2> xy;
        Swz2<ivec2, int, 1, 0, 2> yx;
    };
    ivec2() : x(0), y(0) {}
    ivec2(int a, int b) : x(a), y(b) {}
    explicit ivec2(const vec2 &v) : x((int)v.x), y((int)v.y) {}
    inline explicit ivec2(const uvec2 &u);
    GLSL_VEC_COMMON(ivec2, int, 2)
};
struct uvec2 {
    union {
        struct { uint x, y; };
        uint d[2];
        Swz2<uvec2, uint, 0, 1, 2> xy;
        Swz2<uvec2, uint, 1, 0, 2> yx;
    };
    uvec2() : x(0), y(0) {}
    uvec2(uint a, uint b) : x(a), y(b) {}
    explicit uvec2(uint a) : x(a), y(a) {}
    explicit uvec2(const ivec2 &i) : x((uint)i.x), y((uint)i.y) {}
    GLSL_VEC_COMMON(uvec2, uint, 2)
};
inline vec2::vec2(const uvec2 &u) : x((float)u.x), y((float)u.y) {}
inline vec2::vec2(const ivec2 &i) : x((float)i.x), y((float)i.y) {}
inline ivec2::ivec2(const uvec2 &u) : x((int)u.x), y((int)u.y) {}

struct uvec3 {
    union {
        struct { uint x, y, z; };
        uint d[3];
        Swz2<uvec2, uint, 0, 1, 3> xy;
    };
    uvec3() : x(0), y(0), z(0) {}
    uvec3(uint a, uint b, uint c) : x(a), y(b), z(c) {}
    uvec3(const uvec2 &a, uint c) : x(a.x), y(a.y), z(c) {}
    GLSL_VEC_COMMON(uvec3, uint, 3)
};
struct ivec3 {
    union {
        struct { int x, y, z; };
        int d[3];
        Swz2<ivec2, int, 0, 1, 3> xy;
    };
    ivec3() : x(0), y(0), z(0) {}
    ivec3(int a, int b, int c) : x(a), y(b), z(c) {}
    ivec3(const uvec2 &a, uint c) : x((int)a.x), y((int)a.y), z((int)c) {}
    GLSL_VEC_COMMON(ivec3, int, 3)
};
struct vec3 {
    union {
        struct { float x, y, z; };
        float d[3];
        Swz2<vec2, float, 0, 1, 3> xy;
        Swz2<vec2, float, 0, 2, 3> xz;
        Swz<vec3, float, 3, 0, 1, 2> xyz;
    };
    vec3() : x(0), y(0), z(0) {}
    vec3(float a_, float b_, float c_) : x(a_), y(b_), z(c_) {}
    explicit vec3(float a_) : x(a_), y(a_), z(a_) {}
    vec3(const vec2 &a_, float c_) : x(a_.x), y(a_.y), z(c_) {}
    vec3 &operator+=(const vec3 &o) { x = x + o.x; y = y + o.y; z = z + o.z; return *this; }
    GLSL_VEC_COMMON(vec3, float, 3)
};
struct vec4 {
    union {
        struct { float x, y, z, w; };
        struct { float r, g, b, a; };
        float d[4];
        Swz2<vec2, float, 0, 1, 4> xy;
        Swz2<vec2, float, 2, 3, 4> zw;
        Swz2<vec2, float, 0, 2, 4> xz;
        Swz2<vec2, float, 1, 3, 4> yw;
        Swz2<vec2, float, 0, 3, 4> xw;
        Swz2<vec2, float, 1, 2, 4> yz;
        Swz2<vec2, float, 3, 3, 4> ww;
        Swz<vec3, float, 4, 0, 1, 2> xyz;
        Swz<vec3, float, 4, 0, 1, 3> xyw;
    };
    vec4() : x(0), y(0), z(0), w(0) {}
    vec4(float a_, float b_, float c_, float d_) : x(a_), y(b_), z(c_), w(d_) {}
    vec4(const vec2 &a_, const vec2 &b_) : x(a_.x), y(a_.y), z(b_.x), w(b_.y) {}
    vec4(const vec2 &a_, float c_, float d_) : x(a_.x), y(a_.y), z(c_), w(d_) {}
    GLSL_VEC_COMMON(vec4, float, 4)
};

// ---- operators (component-wise; only the shapes the shaders use) -------------------------------------
inline vec2 operator+(const vec2 &a, const vec2 &b) { return vec2(a.x + b.x, a.y + b.y); }
inline vec2 operator-(const vec2 &a, const vec2 &b) { return vec2(a.x - b.x, a.y - b.y); }
inline vec2 operator*(const vec2 &a, const vec2 &b) { return vec2(a.x * b.x, a.y * b.y); }
inline vec2 operator/(const vec2 &a, const vec2 &b) { return vec2(a.x / b.x, a.y / b.y); }
inline vec2 operator*(const vec2 &a, float s) { return vec2(a.x * s, a.y * s); }
inline vec2 operator*(float s, const vec2 &a) { return vec2(s * a.x, s * a.y); }
inline vec2 operator/(const vec2 &a, float s) { return vec2(a.x / s, a.y / s); }
inline vec2 operator/(float s, const vec2 &a) { return vec2(s / a.x, s / a.y); }
inline vec2 operator+(float s, const vec2 &a) { return vec2(s + a.x, s + a.y); }
inline vec2 operator-(const vec2 &a) { return vec2(-a.x, -a.y); }
// int vector (op) float: GLSL converts the int operand to float first
inline vec2 operator*(const ivec2 &a, float s) { return vec2((float)a.x * s, (float)a.y * s); }
inline vec2 operator-(const ivec2 &a, const vec2 &b) { return vec2((float)a.x - b.x, (float)a.y - b.y); }
inline ivec2 operator+(const ivec2 &a, const ivec2 &b) { return ivec2(a.x + b.x, a.y + b.y); }
inline ivec2 operator-(const ivec2 &a) { return ivec2(-a.x, -a.y); }
inline uvec2 operator>>(const uvec2 &a, int s) { return uvec2(a.x >> s, a.y >> s); }
inline uvec2 operator&(const uvec2 &a, const uvec2 &b) { return uvec2(a.x & b.x, a.y & b.y); }
inline uvec2 operator*(const uvec2 &a, uint s) { return uvec2(a.x * s, a.y * s); }
inline uvec2 operator+(const uvec2 &a, const uvec2 &b) { return uvec2(a.x + b.x, a.y + b.y); }
inline vec4 operator*(const vec4 &a, float s) { return vec4(a.x * s, a.y * s, a.z * s, a.w * s); }
// shapes used by the spatial / particle shaders (water.gdshader, sea_spray_particle.gdshader)
inline vec2 operator+(const vec2 &a, float s) { return vec2(a.x + s, a.y + s); }
inline vec3 operator*(const vec3 &a, const vec3 &b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator*(const vec3 &a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator/(const vec3 &a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
inline vec4 operator/(const vec4 &a, float s) { return vec4(a.x / s, a.y / s, a.z / s, a.w / s); }
inline vec4 operator/(const vec4 &a, const vec4 &b) { return vec4(a.x / b.x, a.y / b.y, a.z / b.z, a.w / b.w); }
inline vec4 operator+(const vec4 &a, const vec4 &b) { return vec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline vec4 operator*(const vec4 &a, const vec4 &b) { return vec4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }

// ---- built-in functions ------------------------------------------------------------------------------
inline float cos(float x) { return ::cosf(x); }
inline float sin(float x) { return ::sinf(x); }
inline float exp(float x) { return ::expf(x); }
inline float log(float x) { return ::logf(x); }
inline float sqrt(float x) { return ::sqrtf(x); }
inline float inversesqrt(float x) { return 1.0f / ::sqrtf(x); }
inline float pow(float x, float y) { return ::powf(x, y); }
inline float pow(float x, int y) { return ::powf(x, (float)y); }
inline float atan(float y, float x) { return ::atan2f(y, x); }
inline float tanh_libm(float x) { return ::tanhf(x); }
inline float tanh_cr(float x) { return (float)::tanh((double)x); }
inline float abs(float x) { return ::fabsf(x); }
inline vec2 abs(const vec2 &v) { return vec2(::fabsf(v.x), ::fabsf(v.y)); }
inline float min(float a, float b) { return b < a ? b : a; }  // GLSL: y < x ? y : x
inline float min(int a, float b) { return min((float)a, b); }
inline float max(float a, float b) { return a < b ? b : a;  }  // GLSL: x < y ? y : x
inline float clamp(float x, float lo, float hi) { return min(max(x, lo), hi); }
inline float mix(float x, float y, float a) { return x * (1.0f - a) + y * a; }
inline float length(const vec2 &v) { return ::sqrtf(v.x * v.x + v.y * v.y); }
inline float length(const vec3 &v) { return ::sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }
inline vec3 normalize(const vec3 &v) { return v / length(v); }  // GLSL: x / length(x)
inline float floor(float x) { return ::floorf(x); }
inline vec2 floor(const vec2 &v) { return vec2(::floorf(v.x), ::floorf(v.y)); }
inline vec2 fract(const vec2 &v) { return vec2(v.x - ::floorf(v.x), v.y - ::floorf(v.y)); }  // GLSL: x - floor(x)
inline vec4 mix(const vec4 &x, const vec4 &y, float a) {
    return vec4(mix(x.x, y.x, a), mix(x.y, y.y, a), mix(x.z, y.z, a), mix(x.w, y.w, a));
}
inline float mod(float x, float y) { return x - y * ::floorf(x / y); }
inline vec2 mod(const ivec2 &x, const ivec2 &y) { return vec2(mod((float)x.x, (float)y.x), mod((float)x.y, (float)y.y)); }
inline uint floatBitsToUint(float f) { uint u; memcpy(&u, &f, 4); return u; }
inline uvec2 floatBitsToUint(const vec2 &v) { return uvec2(floatBitsToUint(v.x), floatBitsToUint(v.y)); }
inline float uintBitsToFloat(uint u) { float f; memcpy(&f, &u, 4); return f; }
inline int findMSB(uint x) { return x ? 31 - __builtin_clz(x) : -1; }

// ---- IEEE half <-> float (RGBA16F storage, round to nearest even) ---------------------------------------
inline uint16_t f32_to_f16(float f) {
    uint32_t x; memcpy(&x, &f, 4);
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
}
inline float f16_to_f32(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu, x;
    if (e == 0) {
        float v = (float)m * 5.9604644775390625e-08f;
        memcpy(&x, &v, 4);
        x |= sign;
    } else if (e == 31) {
        x = sign | 0x7F800000u | (m << 13);
    } else {
        x = sign | ((e + 112u) << 23) | (m << 13);
    }
    float f; memcpy(&f, &x, 4);
    return f;
}

// ---- image2DArray ------------------------------------------------------------------------------------------
struct image2DArray {
    int w = 0, h = 0, layers = 0;
    bool fp32 = false;      // true: R32G32B32A32_SFLOAT storage, false: R16G16B16A16_SFLOAT
    float *f = nullptr;     // fp32 storage
    uint16_t *q = nullptr;  // fp16 storage
};
inline ivec3 imageSize(const image2DArray &im) { return ivec3(im.w, im.h, im.layers); }
inline vec4 imageLoad(const image2DArray &im, const ivec3 &p) {
    size_t t = (((size_t)p.z * im.h + p.y) * im.w + p.x) * 4;
    if (im.fp32) return vec4(im.f[t], im.f[t + 1], im.f[t + 2], im.f[t + 3]);
    return vec4(f16_to_f32(im.q[t]), f16_to_f32(im.q[t + 1]), f16_to_f32(im.q[t + 2]), f16_to_f32(im.q[t + 3]));
}
inline void imageStore(image2DArray &im, const ivec3 &p, const vec4 &v) {
    size_t t = (((size_t)p.z * im.h + p.y) * im.w + p.x) * 4;
    if (im.fp32) {
        for (int i = 0; i < 4; ++i) im.f[t + i] = v.d[i];
    } else {
        for (int i = 0; i < 4; ++i) im.q[t + i] = f32_to_f16(v.d[i]);
    }
}

// ---- sampler2DArray over RGBA16F layers -------------------------------------------------------------------
// texture(): GL_LINEAR minification/magnification, GL_REPEAT wrap, no mipmaps (the maps have one level), as the OpenGL 4.6
// core specification 8.14.2 / Vulkan 16.8 define it: unnormalised coordinate u * size - 0.5, i0 = floor, weights alpha /
// beta = the fractional parts, texel indices wrapped, tau = (1-a)(1-b) t00 + ... evaluated as two lerps along x then one
// along y with exact FP32 weights (hardware quantises the weights to ~8 bits; the reference does not pin that).
struct sampler2DArray {
    int w = 0, h = 0, layers = 0;
    const uint16_t *q = nullptr;  // [layers][h][w][4] FP16 bits
};
inline ivec3 textureSize(const sampler2DArray &s, int) { return ivec3(s.w, s.h, s.layers); }
inline vec4 texture(const sampler2DArray &s, const vec3 &p) {
    int layer = (int)::rintf(p.z);
    layer = layer < 0 ? 0 : (layer >= s.layers ? s.layers - 1 : layer);
    const uint16_t *t = s.q + (size_t)layer * s.w * s.h * 4;
    const float un = p.x * (float)s.w - 0.5f, vn = p.y * (float)s.h - 0.5f;
    const float fi = ::floorf(un), fj = ::floorf(vn);
    const float a = un - fi, b = vn - fj;
    long i0 = (long)fi % s.w, j0 = (long)fj % s.h;
    if (i0 < 0) i0 += s.w;
    if (j0 < 0) j0 += s.h;
    const long i1 = (i0 + 1) % s.w, j1 = (j0 + 1) % s.h;
    vec4 r;
    for (int k = 0; k < 4; ++k) {
        const float t00 = f16_to_f32(t[((size_t)j0 * s.w + i0) * 4 + k]), t10 = f16_to_f32(t[((size_t)j0 * s.w + i1) * 4 + k]);
        const float t01 = f16_to_f32(t[((size_t)j1 * s.w + i0) * 4 + k]), t11 = f16_to_f32(t[((size_t)j1 * s.w + i1) * 4 + k]);
        r.d[k] = (t00 * (1.0f - a) + t10 * a) * (1.0f - b) + (t01 * (1.0f - a) + t11 * a) * b;
    }
    return r;
}

// ---- invocation state + barrier (defined in glsl_ref.cpp) -----------------------------------------------------
extern thread_local uvec3 gl_NumWorkGroups, gl_WorkGroupID, gl_LocalInvocationID, gl_GlobalInvocationID;
void barrier();

static_assert(sizeof(vec2) == 8 && sizeof(vec4) == 16 && sizeof(uvec2) == 8, "buffer element sizes");

}  // namespace glsl

// qualifiers that are meaningless on the CPU
#define shared static
// `in` parameter qualifier (the layout(...) in; line is rewritten by glsl_prep.py before this matters)
#define in
#define GLSL_LOCAL_SIZE(X, Y, Z) static const glsl::uvec3 gl_WorkGroupSize((uint)(X), (uint)(Y), (uint)(Z));
