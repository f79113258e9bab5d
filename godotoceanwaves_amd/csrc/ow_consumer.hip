// ow_consumer.hip -- the read side of the two output arrays, on the device.
//
// What the reference's consumers compute per query point from the RGBA16F layers:
//   water.gdshader:27-39 (vertex)              displacement = sum_i texture(displacements, vec3(UV*scales_i.xy, i)).xyz * scales_i.z
//   water.gdshader:72-82 (fragment)            gradient     = sum_i mix(texture_bicubic, texture, min(1, 0.1 ppm)).xyw * vec3(scales_i.ww, 1)
//                                              (both the bilinear-only sum and the mix with the B-spline filter of :41-68)
//   sea_spray_particle.gdshader:78-96          the spawn mask: unscaled gradient sum -> normal.y window, foam > 0.9
// texture() here is GL_LINEAR + GL_REPEAT on an N x N layer, texel centres at (i + 0.5)/N.  The arithmetic is FP32
// with the weights kept exact (a texture unit quantises them to 8 fractional bits; that is not pinned by the
// reference; the CPU oracle of the test-suite and this kernel both use the exact weights).  This unit is built with
// -ffp-contract=off so that the oracle's restatement of the shaders can be compared with it to the last bit.
#include <hip/hip_runtime.h>

#include "ow_kernels.h"

namespace ow {
namespace {

struct Tap {
    int r0, r1, c0, c1;
    float wx, wy;
};

// texel coordinates and weights of one bilinear lookup at normalised (u, v); u runs along columns
__device__ inline Tap make_tap(float u, float v, int n) {
    const float fx = u * (float)n - 0.5f, fy = v * (float)n - 0.5f;
    const float x0 = floorf(fx), y0 = floorf(fy);
    Tap t;
    t.wx = fx - x0;
    t.wy = fy - y0;
    const int mask = n - 1;  // N is a power of two: two's-complement AND is the positive modulus
    t.c0 = (int)x0 & mask;
    t.c1 = (t.c0 + 1) & mask;
    t.r0 = (int)y0 & mask;
    t.r1 = (t.r0 + 1) & mask;
    return t;
}

__device__ inline void texel_f32(const u16x4 *layer, int n, int r, int c, float out[4]) {
    const u16x4 q = layer[(size_t)r * n + c];
    out[0] = h2f(q.x);
    out[1] = h2f(q.y);
    out[2] = h2f(q.z);
    out[3] = h2f(q.w);
}

__device__ inline void bilinear(const u16x4 *layer, int n, const Tap &t, float out[4]) {
    float a[4], b[4], c[4], d[4];
    texel_f32(layer, n, t.r0, t.c0, a);
    texel_f32(layer, n, t.r0, t.c1, b);
    texel_f32(layer, n, t.r1, t.c0, c);
    texel_f32(layer, n, t.r1, t.c1, d);
    const float ux = 1.0f - t.wx, uy = 1.0f - t.wy;
    for (int k = 0; k < 4; ++k) out[k] = (a[k] * ux + b[k] * t.wx) * uy + (c[k] * ux + d[k] * t.wx) * t.wy;
}

__device__ inline float glsl_mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }

// water.gdshader:41-51 cubic_weights, :53-68 texture_bicubic: cubic B-spline filtering as four bilinear taps
__device__ inline void cubic_weights(float a, float w[4]) {
    const float a2 = a * a, a3 = a2 * a;
    w[0] = (-a3 + a2 * 3.0f - a * 3.0f + 1.0f) / 6.0f;
    w[1] = (a3 * 3.0f - a2 * 6.0f + 4.0f) / 6.0f;
    w[2] = (-a3 * 3.0f + a2 * 3.0f + a * 3.0f + 1.0f) / 6.0f;
    w[3] = a3 / 6.0f;
}
__device__ inline void bicubic(const u16x4 *layer, int n, float u, float v, float out[4]) {
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
    float t_yw[4], t_xw[4], t_yz[4], t_xz[4];
    bilinear(layer, n, make_tap(hx1, hy1, n), t_yw);
    bilinear(layer, n, make_tap(hx0, hy1, n), t_xw);
    bilinear(layer, n, make_tap(hx1, hy0, n), t_yz);
    bilinear(layer, n, make_tap(hx0, hy0, n), t_xz);
    for (int k = 0; k < 4; ++k) out[k] = glsl_mix(glsl_mix(t_yw[k], t_xw[k], wgx), glsl_mix(t_yz[k], t_xz[k], wgx), wgy);
}

__global__ void k_sample_surface(const u16x4 *disp, const u16x4 *norm, int n, int cascades, const float *xz, int count,
                                 SurfaceScales scales, SurfaceSample *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float x = xz[2 * i], z = xz[2 * i + 1];
    float dsum[3] = {0.0f, 0.0f, 0.0f}, g[2] = {0.0f, 0.0f}, gs[2] = {0.0f, 0.0f}, foam = 0.0f;
    float gf[2] = {0.0f, 0.0f}, foam_f = 0.0f;
    const size_t plane = (size_t)n * n;
    for (int c = 0; c < cascades; ++c) {
        const float sx = scales.s[c][0], sy = scales.s[c][1], sz = scales.s[c][2], sw = scales.s[c][3];
        const Tap t = make_tap(x * sx, z * sy, n);
        float d[4], m[4];
        bilinear(disp + c * plane, n, t, d);
        bilinear(norm + c * plane, n, t, m);
        for (int k = 0; k < 3; ++k) dsum[k] += d[k] * sz;
        g[0] += m[0];
        g[1] += m[1];
        gs[0] += m[0] * sw;
        gs[1] += m[1] * sw;
        foam += m[3];
        {   // water.gdshader:74-82 fragment(): bicubic and bilinear mixed by the pixels per metre of this cascade
            float bc[4];
            const float ppm = (float)n * fminf(sx, sy);
            const float a = fminf(1.0f, ppm * 0.1f);
            bicubic(norm + c * plane, n, x * sx, z * sy, bc);
            gf[0] += glsl_mix(bc[0], m[0], a) * sw;
            gf[1] += glsl_mix(bc[1], m[1], a) * sw;
            foam_f += glsl_mix(bc[3], m[3], a) * 1.0f;
        }
    }
    // sea_spray_particle.gdshader:83-89
    const float normal_y = 1.0f / sqrtf(g[0] * g[0] + 1.0f + g[1] * g[1]);
    const float normal_factor = glsl_mix(0.25f, 1.0f, fminf((normal_y - 0.92f) / (0.99f - 0.92f), 1.0f));
    const float foam_factor = glsl_mix(0.25f, 1.0f, fminf((foam - 0.9f) / (1.0f - 0.9f), 1.0f));
    SurfaceSample s;
    s.displacement[0] = dsum[0];
    s.displacement[1] = dsum[1];
    s.displacement[2] = dsum[2];
    s.gradient[0] = g[0];
    s.gradient[1] = g[1];
    s.gradient_scaled[0] = gs[0];
    s.gradient_scaled[1] = gs[1];
    s.foam = foam;
    s.normal_factor = normal_factor;
    s.foam_factor = foam_factor;
    s.scale_factor = normal_factor * foam_factor;
    s.spray_active = (normal_factor >= 0.0f && normal_factor <= 1.0f && foam > 0.9f) ? 1 : 0;
    s.gradient_fragment[0] = gf[0];
    s.gradient_fragment[1] = gf[1];
    s.foam_fragment = foam_f;
    s.reserved = 0.0f;
    out[i] = s;
}

}  // namespace

hipError_t launch_sample_surface(int n, int cascades, const DeviceBuffers &buf, const float *xz_dev, int count,
                                 const SurfaceScales &scales, SurfaceSample *out_dev, hipStream_t s) {
    if (count <= 0) return hipSuccess;
    const int threads = 256;
    hipLaunchKernelGGL(k_sample_surface, dim3((count + threads - 1) / threads), dim3(threads), 0, s, buf.disp, buf.norm, n, cascades,
                       xz_dev, count, scales, out_dev);
    return hipGetLastError();
}

}  // namespace ow
