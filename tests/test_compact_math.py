"""The algebra behind the compact intermediate (DESIGN.md "compact intermediate"), in NumPy FP64 against the independent
FP64 twin of the reference pipeline (tests/np_twin.py).

The reference transforms four packed complex layers (spectrum_modulate.glsl:72-89).  Away from the two Nyquist lines
(texel column id.x = 0 and texel row id.y = 0) all eight fields are real, and three of them are derivatives, along the
axis the SECOND row pass transforms, of three others:
    dhx_dx = i ky hx,   dhy_dx = i ky hy,   dhz_dx = i ky hz        (spectrum_modulate.glsl:72-82)
so only five real fields have to cross the intermediate: T0 = hx + i hy, T1 = hz (alone: Hermitian along ky, which is
what lets (1 - ky) T1 carry hz + i dhz_dx through ONE transform, and why only the rows ky >= 0 of T1 are kept),
T2 = dhy_dz + i dhz_dz -- two and a half layers instead of four.
On the Nyquist lines the reference's packed layers are not Hermitian-consistent (odd multipliers meet an un-mirrored
wave number) and leak a few % into the other component; those lines are handled in closed form:
    column id.x = 0:  T1 gets 0, T2 gets ux (ky - i kx) h, and pass 2 adds P(ky) = (kx + i ux) h to the derived i ky T0
    row    id.y = 0:  three extra row transforms Q1..Q3 whose results replace element ky-index 0 of the pass-2 inputs
This test holds the closed forms to the reference's outputs at 1e-12 (they agree to rounding, 1e-16)."""
import numpy as np
import pytest

import np_twin as T


def compact_pipeline(n, tile, t, h0, h0m):
    idy, idx = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    kx = (idx - n * 0.5) * 2 * np.pi / tile[0]
    ky = (idy - n * 0.5) * 2 * np.pi / tile[1]
    k = np.hypot(kx, ky) + 1e-6
    ux, uy = kx / k, ky / k
    m = np.exp(1j * np.sqrt(T.G * k * np.tanh(k * 20.0)) * t)
    h = h0 * m + h0m * np.conj(m)
    # ---- pass 1: three layers, column id.x = 0 in closed form ----
    Z0, Z1, Z2 = 1j * (1 + uy) * h, 1j * ux * h, 1j * kx * (1 - ux) * h
    col = idx == 0
    Z1 = np.where(col, 0, Z1)
    Z2 = np.where(col, ux * (ky - 1j * kx) * h, Z2)
    P = ((kx + 1j * ux) * h)[:, 0]
    Tm = [np.fft.ifft(Z, axis=1) * n for Z in (Z0, Z1, Z2)]
    # ---- row id.y = 0: three extra transforms (the fourth, Q0, is T0's own row) ----
    Q1, Q2, Q3 = -ky[0] * uy[0] * h[0], (1j * ux[0] - ky[0]) * h[0], (1j * kx[0] * (1 - ux[0]) + ky[0] * ux[0]) * h[0]
    Q1[0] = ((-ky * uy + kx + 1j * ux) * h)[0, 0]        # the corner texel mirrors onto itself
    Q2[0] = (-ky * (1 + 1j * ux) * h)[0, 0]
    Q3[0] = (-1j * kx * ux * h)[0, 0]
    R = [None] + [np.fft.ifft(q) * n for q in (Q1, Q2, Q3)]
    # ---- pass 2: four transforms per row x' ----
    kyv = ky[:, 0]
    # only the rows y >= N/2 of T1 are kept; row N - y is the conjugate (T1 = hz alone is Hermitian along the second axis)
    S = Tm[1].copy()
    S[1:n // 2] = np.nan
    c1 = S.copy()
    c1[1:n // 2] = np.conj(S[n - np.arange(1, n // 2)])
    out = np.zeros((4, n, n), complex)
    for xp in range(n):
        G = [Tm[0][:, xp].copy(), 1j * kyv * Tm[0][:, xp] + P, (1 - kyv) * c1[:, xp], Tm[2][:, xp].copy()]
        for j in (1, 2, 3):
            G[j][0] = R[j][xp]
        F = [np.fft.ifft(g) * n for g in G]
        out[0, xp] = F[0]                                # hx + i hy
        out[1, xp] = F[2].real + 1j * F[1].imag          # hz + i dhy_dx
        out[2, xp] = F[3].real + 1j * F[1].real          # dhy_dz + i dhx_dx
        out[3, xp] = F[3].imag + 1j * F[2].imag          # dhz_dz + i dhz_dx
    return out, Tm


@pytest.mark.parametrize("n,tile", [(32, (88.0, 88.0)), (64, (88.0, 57.0)), (128, (16.0, 250.0))])
def test_three_layer_intermediate_reproduces_the_four_layer_reference(n, tile):
    p = dict(tile_length=tile, depth=20.0, peak_frequency=T.jonswap_peak(10, 150e3), alpha=T.jonswap_alpha(10, 150e3),
             wind_speed=10.0, angle=0.35, swell=0.8, detail=1.0, spread=0.2, seed=(1000, -2000))
    h0, h0m = T.spectrum(n, p)
    ref = T.ifft2_ref(T.modulate(n, tile, 20.0, 123.4, h0, h0m))
    out, Tm = compact_pipeline(n, tile, 123.4, h0, h0m)
    assert np.abs(out - ref).max() / np.abs(ref).max() < 1e-12
    # T1 is Hermitian along the second axis (which is why only half of it is stored)
    assert np.abs(Tm[1][1:] - np.conj(Tm[1][1:][::-1])).max() < 1e-12 * np.abs(Tm[1]).max()


def test_the_nyquist_lines_matter():
    """dropping the closed-form line handling changes the derivative channels by far more than the parity tolerance"""
    n, tile = 64, (88.0, 57.0)
    p = dict(tile_length=tile, depth=20.0, peak_frequency=T.jonswap_peak(10, 150e3), alpha=T.jonswap_alpha(10, 150e3),
             wind_speed=10.0, angle=0.35, swell=0.8, detail=1.0, spread=0.2, seed=(1000, -2000))
    h0, h0m = T.spectrum(n, p)
    X = T.modulate(n, tile, 20.0, 123.4, h0, h0m)
    ref = T.ifft2_ref(X)
    X[:, 0, :] = 0
    X[:, :, 0] = 0
    assert np.abs(T.ifft2_ref(X) - ref).max() / np.abs(ref).max() > 1e-3
