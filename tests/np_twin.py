"""Independent NumPy FP64 "truth" twin of the reference pipeline (test infrastructure).

Written from the math of the reference shaders (vectorised, FP64, np.fft for the transform),
NOT from oracle/ow_oracle.c, so that two independent restatements have to agree:
  spectrum_compute.glsl, spectrum_modulate.glsl, wave_generator.gd:77-82 (rows->transpose->rows
  == (N^2 * ifft2(X))^T), fft_unpack.glsl.
Only the integer hash and the uniform u = float32(n)/float32(2^31) are kept in their native
widths (they define the Gaussian draw); everything after is FP64.
"""
import numpy as np

G = 9.81
PI = np.pi


def hash_uniform(ix, iy):
    """spectrum_compute.glsl:34-41 on uint32 arrays -> two FP64 uniforms in [0,1]."""
    x = ix.astype(np.uint32)
    y = iy.astype(np.uint32)
    with np.errstate(over="ignore"):
        h = y + np.uint32(374761393) + x * np.uint32(3266489917)
        h = np.uint32(2246822519) * (h ^ (h >> np.uint32(15)))
        h = np.uint32(3266489917) * (h ^ (h >> np.uint32(13)))
        n = h ^ (h >> np.uint32(16))
        n2 = n * np.uint32(48271)
    den = np.float32(0x7FFFFFFF)
    u1 = (((n >> np.uint32(1)) & np.uint32(0x7FFFFFFF)).astype(np.float32) / den).astype(np.float64)
    u2 = (((n2 >> np.uint32(1)) & np.uint32(0x7FFFFFFF)).astype(np.float32) / den).astype(np.float64)
    return u1, u2


def jonswap_alpha(U, F):
    return 0.076 * (U ** 2 / (F * G)) ** 0.22


def jonswap_peak(U, F):
    return 22.0 * (G * G / (U * F)) ** (1.0 / 3.0)


def _amplitude(idx, idy, n, p):
    """get_spectrum_amplitude (spectrum_compute.glsl:103-115), complex128."""
    Lx, Ly = p["tile_length"]
    dkx, dky = 2 * PI / Lx, 2 * PI / Ly
    kx = (idx - n * 0.5) * dkx
    ky = (idy - n * 0.5) * dky
    k = np.hypot(kx, ky) + 1e-6
    theta = np.arctan2(kx, ky)
    depth = p["depth"]
    a = k * depth
    b = np.tanh(a)
    w = np.sqrt(G * k * b)
    dw = 0.5 * G * (b + a * (1 - b * b)) / w
    w_norm = dw / k * dkx * dky
    w_p, alpha = p["peak_frequency"], p["alpha"]
    with np.errstate(over="ignore", divide="ignore", invalid="ignore"):
        sigma = np.where(w <= w_p, 0.07, 0.09)
        r = np.exp(-(w - w_p) ** 2 / (2 * sigma ** 2 * w_p ** 2))
        jons = alpha * G * G / w ** 5 * np.exp(-1.25 * (w_p / w) ** 4) * 3.3 ** r
        w_h = np.minimum(w * np.sqrt(depth / G), 2.0)
        kit = np.where(w_h <= 1.0, 0.5 * w_h ** 2, 1.0 - 0.5 * (2.0 - w_h) ** 2)
        S = jons * kit
        pr = w / w_p
        s = np.where(w <= w_p, 6.97 * np.abs(pr) ** 4.06,
                     9.77 * np.abs(pr) ** (-2.33 - 1.45 * (p["wind_speed"] * w_p / G - 1.17)))
        s = s + 16.0 * np.tanh(w_p / w) * p["swell"] ** 2
        sq = np.sqrt(s)
        norm = np.where(s < 0.4, 0.5 / PI + s * (0.220636 + s * (-0.109 + s * 0.090)),
                        (1 / np.sqrt(PI)) * (sq * 0.5 + (1 / sq) * 0.0625))
        D = norm * np.abs(np.cos((theta - p["angle"]) * 0.5)) ** (2 * s)
        d = ((0.5 / PI) * p["spread"] + D * (1 - p["spread"])) * np.exp(-(1 - p["detail"]) ** 2 * k * k)
    u1, u2 = hash_uniform((idx + p["seed"][0]).astype(np.int64), (idy + p["seed"][1]).astype(np.int64))
    with np.errstate(divide="ignore"):
        rr = np.sqrt(-2.0 * np.log(u1))
    th = 2 * PI * u2
    gauss = rr * np.cos(th) + 1j * rr * np.sin(th)
    return gauss * np.sqrt(2 * S * d * w_norm)


def spectrum(n, p):
    """returns (h0(k), conj(h0(-k))) as two complex128 [y][x] arrays (spectrum_compute.glsl:117-125)."""
    idy, idx = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    h0 = _amplitude(idx, idy, n, p)
    h0m = np.conj(_amplitude((-idx) % n, (-idy) % n, n, p))
    return h0, h0m


def modulate(n, tile, depth, t, h0, h0m, omega=None):
    """spectrum_modulate.glsl:53-90 -> 4 packed complex layers [layer][y][x].
    `omega` (optional) overrides the FP64 dispersion with a given array (e.g. the FP32 one)."""
    idy, idx = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    kx = (idx - n * 0.5) * 2 * PI / tile[0]
    ky = (idy - n * 0.5) * 2 * PI / tile[1]
    k = np.hypot(kx, ky) + 1e-6
    ux, uy = kx / k, ky / k
    w = np.sqrt(G * k * np.tanh(k * depth)) if omega is None else omega.astype(np.float64)
    ph = w * t
    m = np.exp(1j * ph)
    h = h0 * m + h0m * np.conj(m)
    hi = 1j * h
    hx, hy, hz = hi * uy, h, hi * ux
    dhy_dx, dhy_dz = hi * ky, hi * kx
    dhx_dx, dhz_dz, dhz_dx = -h * ky * uy, -h * kx * ux, -h * ky * ux
    return np.stack([hx + 1j * hy, hz + 1j * dhy_dx, dhy_dz + 1j * dhx_dx, dhz_dz + 1j * dhz_dx])


def ifft2_ref(layers):
    """wave_generator.gd:77-82: rows -> transpose -> rows, no 1/N, no second transpose."""
    n = layers.shape[-1]
    return np.transpose(np.fft.ifft2(layers, axes=(-2, -1)) * (n * n), (0, 2, 1))


def unpack(out, whitecap, grow, decay, foam_prev=None):
    """fft_unpack.glsl:33-70 in FP64 -> dict of channels [y][x] (no FP16 rounding)."""
    n = out.shape[-1]
    iy, ix = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    sign = 1.0 - 2.0 * ((ix & 1) ^ (iy & 1))
    o = out * sign
    hx, hy, hz = o[0].real, o[0].imag, o[1].real
    dhy_dx, dhy_dz, dhx_dx = o[1].imag, o[2].real, o[2].imag
    dhz_dz, dhz_dx = o[3].real, o[3].imag
    J = (1 + dhx_dx) * (1 + dhz_dz) - dhz_dx ** 2
    foam = np.zeros_like(J) if foam_prev is None else foam_prev.astype(np.float64)
    foam = np.clip(foam * np.exp(-decay) + np.maximum(0.0, whitecap - J) * grow, 0.0, 1.0)
    gx = dhy_dx / (1 + np.abs(dhx_dx))
    gy = dhy_dz / (1 + np.abs(dhz_dz))
    return np.stack([hx, hy, hz, gx, gy, dhx_dx, foam, J], axis=-1)
