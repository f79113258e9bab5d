"""Consumer-side view of the two output textures, as NumPy: how water.gdshader samples them and
what the channels mean physically.  Test infrastructure: validates orientation, tiling and scale of the maps end to
end, independently of how the transform was computed.

water.gdshader:27-39  vertex():   UV = VERTEX.xz;  displacement += texture(displacements, vec3(UV*scales.xy, i)).xyz * scales.z
=> texture coordinate u (pixel COLUMN) runs along world x, v (pixel ROW) along world z, one tile = tile_length metres,
   repeat addressing.  Because the generator skips the second transpose (wave_generator.gd:77-82), pixel column is the
   spatial index conjugate to k_vec.y and pixel row the one conjugate to k_vec.x -- which is why spectrum_modulate.glsl
   pairs the x-displacement and the x-derivatives with k.y (:72-82).
"""
import numpy as np


def texture_bilinear(img, u, v):
    """GL_LINEAR + GL_REPEAT lookup of img[row, col, channel] at normalised (u, v); texel centres at (i + 0.5) / N"""
    n_rows, n_cols = img.shape[:2]
    x, y = np.asarray(u) * n_cols - 0.5, np.asarray(v) * n_rows - 0.5
    x0, y0 = np.floor(x).astype(int), np.floor(y).astype(int)
    fx, fy = (x - x0)[..., None], (y - y0)[..., None]
    c0, c1, r0, r1 = x0 % n_cols, (x0 + 1) % n_cols, y0 % n_rows, (y0 + 1) % n_rows
    return (img[r0, c0] * (1 - fx) + img[r0, c1] * fx) * (1 - fy) + (img[r1, c0] * (1 - fx) + img[r1, c1] * fx) * fy


def displacement_at(disp_maps, map_scales, world_x, world_z):
    """water.gdshader:31-37: sum over cascades of texture(displacements, vec3(UV*scales.xy, i)).xyz * scales.z"""
    out = 0.0
    for img, (sx, sy, sz, _) in zip(disp_maps, map_scales):
        out = out + texture_bilinear(np.asarray(img, np.float64), world_x * sx, world_z * sy)[..., :3] * sz
    return out


def d_dx_world(field, tile_length_x):
    """Spectral derivative of a periodic map along WORLD X = along pixel columns (axis 1)."""
    n = field.shape[1]
    k = 2.0 * np.pi * np.fft.fftfreq(n, d=tile_length_x / n)
    return np.real(np.fft.ifft(np.fft.fft(field.astype(np.float64), axis=1) * (1j * k)[None, :], axis=1))


def d_dz_world(field, tile_length_y):
    n = field.shape[0]
    k = 2.0 * np.pi * np.fft.fftfreq(n, d=tile_length_y / n)
    return np.real(np.fft.ifft(np.fft.fft(field.astype(np.float64), axis=0) * (1j * k)[:, None], axis=0))


def texture_bspline(img, u, v):
    """Cubic B-spline filtering of img[row, col, channel] at normalised (u, v), written directly as the 4 x 4-tap weighted sum
    (water.gdshader:53-68 gets the same value from four bilinear taps, GPU Gems 2 ch. 20); repeat addressing"""
    n_rows, n_cols = img.shape[:2]
    x, y = np.asarray(u, np.float64) * n_cols - 0.5, np.asarray(v, np.float64) * n_rows - 0.5
    x0, y0 = np.floor(x).astype(int), np.floor(y).astype(int)
    fx, fy = x - x0, y - y0

    def weights(a):
        return [(-a ** 3 + 3 * a ** 2 - 3 * a + 1) / 6, (3 * a ** 3 - 6 * a ** 2 + 4) / 6, (-3 * a ** 3 + 3 * a ** 2 + 3 * a + 1) / 6, a ** 3 / 6]

    wx, wy = weights(fx), weights(fy)
    out = 0.0
    for b in range(4):
        for a in range(4):
            out = out + (wx[a] * wy[b])[..., None] * img[(y0 - 1 + b) % n_rows, (x0 - 1 + a) % n_cols]
    return out


def gradient_fragment_at(norm_maps, map_scales, world_x, world_z):
    """water.gdshader:74-82: sum over cascades of mix(bicubic, bilinear, min(1, 0.1 * map_size * min(scales.xy))).xyw * (scales.ww, 1)"""
    out = 0.0
    for img, (sx, sy, _, sw) in zip(norm_maps, map_scales):
        img = np.asarray(img, np.float64)
        a = min(1.0, 0.1 * img.shape[0] * min(sx, sy))
        val = texture_bspline(img, world_x * sx, world_z * sy) * (1 - a) + texture_bilinear(img, world_x * sx, world_z * sy) * a
        out = out + val[..., [0, 1, 3]] * np.array([sw, sw, 1.0])
    return out
