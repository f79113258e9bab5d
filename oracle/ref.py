"""ctypes binding of oracle/_ref/libglsl_ref.so = the REFERENCE'S OWN compute shaders compiled as C++
(oracle/glsl_ref.cpp + oracle/glsl_shim.h over the .glsl sources where they lie in the reference checkout).
TEST INFRASTRUCTURE ONLY: used to pin oracle/ow_oracle.c and to generate tests/golden/ (tests/golden/make_golden.py).

The library can only be BUILT where /root/reference exists (oracle/Makefile target `ref`); the built .so is
git-ignored but travels with the working tree, and `available()` says whether it is there.
"""
import ctypes as C
import math
import os
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libglsl_ref.so")
_lib = None


class RefSpectrumPC(C.Structure):
    _fields_ = [("seed", C.c_int32 * 2), ("tile_length", C.c_float * 2), ("alpha", C.c_float),
                ("peak_frequency", C.c_float), ("wind_speed", C.c_float), ("angle", C.c_float),
                ("depth", C.c_float), ("swell", C.c_float), ("detail", C.c_float), ("spread", C.c_float)]


def available():
    return os.path.exists(_PATH)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_PATH)
        f32p, u16p = np.ctypeslib.ndpointer(np.float32, flags="C"), np.ctypeslib.ndpointer(np.uint16, flags="C")
        L.ref_spectrum_compute.argtypes = [C.c_int, C.POINTER(RefSpectrumPC), f32p]
        L.ref_spectrum_modulate.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, f32p, f32p]
        L.ref_fft_butterfly.argtypes = [C.c_int, f32p]
        L.ref_fft_compute.argtypes = [C.c_int, f32p, f32p]
        L.ref_transpose.argtypes = [C.c_int, f32p, f32p]
        L.ref_fft_unpack.argtypes = [C.c_int, f32p, C.c_float, C.c_float, C.c_float, u16p, u16p]
        L.ref_sample_surface.argtypes = [C.c_int, C.c_int, u16p, u16p, f32p, f32p, C.c_int, C.c_void_p, f32p]
        _lib = L
    return _lib


G = 9.81  # wave_generator.gd:5


def jonswap_alpha(U, F):  # wave_generator.gd:116-117 (GDScript float = FP64)
    return 0.076 * math.pow(U * U / (F * G), 0.22)


def jonswap_peak(U, F):  # wave_generator.gd:120-121
    return 22.0 * math.pow(G * G / (U * F), 1.0 / 3.0)


class RefCascade:
    """One cascade of WaveGenerator (wave_generator.gd:17-109) driven dispatch by dispatch through the
    reference's shaders: init_gpu allocations (:31-35), fft_butterfly once (:52-54), then per update
    spectrum_compute if dirty (:68-72), spectrum_modulate (:73), fft_compute, transpose, fft_compute
    (:79-82), fft_unpack (:85)."""

    DEPTH = 20.0  # wave_generator.gd:6

    def __init__(self, n, preset):
        self.L, self.n, self.p = lib(), n, dict(preset)
        stages = int(round(math.log2(n)))
        self.spectrum = np.zeros((n, n, 4), np.float32)
        self.butterfly = np.zeros((stages, n, 4), np.float32)
        self.fft = np.zeros((2, 4, n, n, 2), np.float32)          # fft_buffer: two halves of 4 layers
        self.displacement = np.zeros((n, n, 4), np.uint16)
        self.normal = np.zeros((n, n, 4), np.uint16)               # undefined in Vulkan; zero here
        self.time = float(preset["time"])
        self.dirty = True
        self.L.ref_fft_butterfly(n, self.butterfly)

    def update(self, delta):
        p, n, L = self.p, self.n, self.L
        # update(): wave_generator.gd:101-106
        self.time += delta
        grow = delta * p["foam_amount"] * 7.5
        decay = delta * max(0.5, 10.0 - p["foam_amount"]) * 1.15
        # _update(): :65-85
        if self.dirty:
            pc = RefSpectrumPC()
            F = p["fetch_length"] * 1e3
            pc.seed[0], pc.seed[1] = p["spectrum_seed"]
            pc.tile_length[0], pc.tile_length[1] = p["tile_length"]
            pc.alpha, pc.peak_frequency = jonswap_alpha(p["wind_speed"], F), jonswap_peak(p["wind_speed"], F)
            pc.wind_speed, pc.angle, pc.depth = p["wind_speed"], math.radians(p["wind_direction"]), self.DEPTH
            pc.swell, pc.detail, pc.spread = p["swell"], p["detail"], p["spread"]
            L.ref_spectrum_compute(n, C.byref(pc), self.spectrum)
            self.dirty = False
        flat = self.fft.reshape(-1)
        t0 = time.perf_counter()                                   # (bench.py's cpu_baseline.reference_shaders: the steady-state dispatches of one update)
        L.ref_spectrum_modulate(n, p["tile_length"][0], p["tile_length"][1], self.DEPTH, self.time, self.spectrum, flat)
        L.ref_fft_compute(n, self.butterfly, flat)
        L.ref_transpose(n, self.butterfly, flat)
        self.intermediate = self.fft[0].copy()                     # half 0 after the transpose
        L.ref_fft_compute(n, self.butterfly, flat)
        L.ref_fft_unpack(n, flat, p["whitecap"], grow, decay, self.displacement, self.normal)
        self.last_steady_s = time.perf_counter() - t0


def sample_surface(displacements, normals, map_scales, world_xz):
    """The reference's own consumer statements (water.gdshader:27-37,41-82, sea_spray_particle.gdshader:80-89,103-107, copied out
    of the .gdshader files at build time by glsl_prep.py --extract) evaluated at world points.  Returns (records, displacement as
    the particle shader sums it); records use oracle.SURFACE_SAMPLE's layout, `gradient_scaled` stays 0 (no statement of its
    own in the reference)."""
    from . import oracle as O
    d = np.ascontiguousarray(np.asarray(displacements).view(np.uint16))
    m = np.ascontiguousarray(np.asarray(normals).view(np.uint16))
    sc = np.ascontiguousarray(map_scales, np.float32)
    xz = np.ascontiguousarray(world_xz, np.float32)
    out = np.zeros(len(xz), O.SURFACE_SAMPLE)
    dp = np.zeros((len(xz), 3), np.float32)
    lib().ref_sample_surface(d.shape[1], len(sc), d, m, sc, xz, len(xz), out.ctypes.data, dp)
    return out, dp
