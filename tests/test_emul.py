"""CPU emulation of the lane-level device code (godotoceanwaves_amd/csrc/ow_device.h compiled as plain
C++, 64 lanes stepped phase by phase) against NumPy and the oracle.  Guards the index math, twiddles,
layouts and orientation of the HIP kernels on a machine without a GPU.  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import helpers as H
from godotoceanwaves_amd.presets import DEPTH, UPDATE_DELTA, cascade_preset
from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "godotoceanwaves_amd", "csrc")


class PC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("seed_x", "seed_y")] + \
               [(n, C.c_float) for n in ("tile_x", "tile_y", "alpha", "peak_frequency", "wind_speed", "angle", "depth", "swell", "detail", "spread")]


class CF(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("tile_x", "tile_y", "time", "whitecap", "foam_grow_rate", "foam_decay")] + \
               [("cascade", C.c_int32), ("pad0", C.c_int32)]


@pytest.fixture(scope="module")
def emul():
    so = os.path.join(HERE, "emul", "libemul.so")
    srcs = [os.path.join(HERE, "emul", "emul.cpp")] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas",
                        "-I", CSRC, srcs[0], "-o", so], check=True)
    E = C.CDLL(so)
    f32p = np.ctypeslib.ndpointer(np.float32, flags="C")
    u16p = np.ctypeslib.ndpointer(np.uint16, flags="C")
    E.emul_rows_fft.argtypes = [C.c_int, f32p, f32p, C.c_int]
    E.emul_rows_fft_half_table.argtypes = [f32p, f32p, C.c_int]
    E.emul_spectrum.argtypes = [C.c_int, C.POINTER(PC), f32p, f32p, f32p]
    E.emul_spectrum_fast.argtypes = [C.c_int, C.POINTER(PC), f32p]
    E.emul_frame.argtypes = [C.c_int, f32p, f32p, C.POINTER(CF), f32p, u16p, u16p, u16p, f32p]
    E.emul_frame_compact.argtypes = [C.c_int, f32p, f32p, C.POINTER(CF), f32p, u16p, u16p, u16p, f32p]
    E.emul_frame_lp.argtypes = [C.c_int, C.c_int, f32p, f32p, C.POINTER(CF), f32p, u16p, u16p, u16p, f32p]
    E.emul_sincos.argtypes = [C.c_int, f32p, f32p, f32p]
    return E


def _spectrum_cases():
    from edge_presets import edge_presets
    return [(f"preset{i}", cascade_preset(i)) for i in range(8)] + sorted(edge_presets().items())


@pytest.mark.parametrize("name,p", _spectrum_cases(), ids=[k for k, _ in _spectrum_cases()])
def test_the_kernels_cheaper_amplitude_stays_within_the_parity_budget(emul, name, p):
    """k_spectrum evaluates the reference's formulas in a cheaper form (ow_device.h spectrum_amplitude_fast: integer powers by multiplication, the
    three other powf as exp2(e log2 x), Cody-Waite sincos, reciprocals for the divisions, tanh saturated to 1 where it rounds to 1).  On the CPU
    build -- the same code over glibc's exp2f / log2f -- its h0 stays within a QUARTER of the 2e-5 the GPU parity tests allow against the oracle
    (tests/test_gpu_parity.py test_spectrum_and_omega), over the eight presets and every range-end preset; the GPU's 1-ulp v_exp_f32 / v_log_f32 add
    an ulp each, amplified by the same exponents.  The literal form (spectrum_amplitude) stays bit-equal to the oracle: test_frame_matches_oracle."""
    n = 256
    pc = H.spectrum_pc(p)
    epc = PC(p["spectrum_seed"][0], p["spectrum_seed"][1], p["tile_length"][0], p["tile_length"][1], pc.alpha, pc.peak_frequency,
             pc.wind_speed, pc.angle, DEPTH, p["swell"], p["detail"], p["spread"])
    fast = np.zeros((n, n, 2), np.float32)
    emul.emul_spectrum_fast(n, C.byref(epc), fast)
    ref = O.spectrum_compute(n, pc)[..., :2]
    assert np.isfinite(fast).all()
    err = H.relmax(fast, ref)
    print(f"{name}: h0 of the cheaper form vs the oracle {err:.2e}")
    assert err < 5e-6


def test_sincos_phase_accuracy(emul):
    """phase arguments up to 2e4 rad (omega ~ 70 rad/s x t ~ 160 s): |err| <= 2e-7 (1-2 ulp at 1.0)"""
    rng = np.random.default_rng(1)
    ph = (rng.random(400000) * 2e4).astype(np.float32)
    sn, cs = np.zeros_like(ph), np.zeros_like(ph)
    emul.emul_sincos(len(ph), ph, sn, cs)
    assert np.abs(sn - np.sin(ph.astype(np.float64))).max() < 2e-7
    assert np.abs(cs - np.cos(ph.astype(np.float64))).max() < 2e-7


@pytest.mark.parametrize("n", [128, 256, 512, 1024, 2048])
def test_row_ifft_is_unnormalised_inverse_dft(emul, n):
    """fft_compute.glsl row pass == N * ifft; asymmetric random input catches index swaps"""
    rng = np.random.default_rng(n)
    rows = 8
    x = rng.standard_normal((rows, n, 2)).astype(np.float32)
    y = np.zeros_like(x)
    assert emul.emul_rows_fft(n, x, y, rows) == 0
    ref = np.fft.ifft(x[..., 0].astype(np.float64) + 1j * x[..., 1], axis=1) * n
    assert H.relmax(y[..., 0] + 1j * y[..., 1], ref) < 5e-7


def test_half_twiddle_table_of_the_2048_pass_2(emul):
    """ow_device.h "HALF TABLE": the row's second wave takes the first wave's stage-0 twiddles times exp(2 pi i k / 32) instead of its own
    half of a 16 KB table (so that a 4-column block of pass 2 fits a CU twice).  Still the unnormalised inverse DFT, to the same accuracy,
    and within a few FP32 ulps of the full-table transform."""
    rng = np.random.default_rng(5)
    n, rows = 2048, 8
    x = rng.standard_normal((rows, n, 2)).astype(np.float32)
    y, full = np.zeros_like(x), np.zeros_like(x)
    assert emul.emul_rows_fft_half_table(x, y, rows) == 0 and emul.emul_rows_fft(n, x, full, rows) == 0
    ref = np.fft.ifft(x[..., 0].astype(np.float64) + 1j * x[..., 1], axis=1) * n
    assert H.relmax(y[..., 0] + 1j * y[..., 1], ref) < 5e-7
    assert H.relmax(y, full) < 3e-7 and not np.array_equal(y, full)   # a different rounding of half the stage-0 twiddles, nothing more


@pytest.mark.parametrize("intermediate", ["reference_layout", "compact", "reference_layout_lp", "compact_lp"])
@pytest.mark.parametrize("n,ci", [(128, 0), (256, 2), (512, 1), (128, "non_square_tile"), (128, "late_time")])
def test_emulated_kernels_match_oracle(emul, n, ci, intermediate):
    """every kernel family: the four-layer intermediate of the reference and the compact one (Pass1::layer_input_c,
    tests/test_compact_math.py), each with the standard pass 2 and with the layer-parallel one (four lane groups per row,
    Pass2::unpack_texel)"""
    frame_fn = {"reference_layout": emul.emul_frame, "compact": emul.emul_frame_compact,
                "reference_layout_lp": lambda n_, *a: emul.emul_frame_lp(n_, 0, *a),
                "compact_lp": lambda n_, *a: emul.emul_frame_lp(n_, 1, *a)}[intermediate]
    if intermediate.startswith("compact") and n < 256:
        n = 256  # the compact lane code needs N/16 to be a multiple of the 16-row line of T (the kernels exist for N >= 1024 only)
    if isinstance(ci, str):
        from edge_presets import edge_presets
        p = edge_presets()[ci]
    else:
        p = cascade_preset(ci)
    pc = H.spectrum_pc(p)
    epc = PC(p["spectrum_seed"][0], p["spectrum_seed"][1], p["tile_length"][0], p["tile_length"][1], pc.alpha, pc.peak_frequency,
             pc.wind_speed, pc.angle, DEPTH, p["swell"], p["detail"], p["spread"])
    h0, h0a, om = np.zeros((n, n, 4), np.float32), np.zeros((n, n, 2), np.float32), np.zeros((n, n), np.float32)
    emul.emul_spectrum(n, C.byref(epc), h0, h0a, om)
    # same libm on both sides + contraction off => the one-time spectrum and omega are BIT-identical
    assert np.array_equal(h0, O.spectrum_compute(n, pc))
    assert np.array_equal(om, O.omega(n, p["tile_length"], DEPTH))

    g = O.Generator(n, 1, DEPTH)
    H.set_params(g.params[0], p)
    norm = np.zeros((n, n, 4), np.uint16)
    foam = np.zeros((n * n,), np.uint16)  # the context's private FP16 foam plane (device order)
    for frame in range(3):
        g.update_all(UPDATE_DELTA)
        P = g.params[0]
        cf = CF(p["tile_length"][0], p["tile_length"][1], np.float32(P.time), p["whitecap"], np.float32(P.foam_grow_rate),
                np.exp(-np.float32(P.foam_decay_rate), dtype=np.float32), 0, 0)
        T = np.zeros((n * n * 4 * 2,), np.float32)
        disp, f32 = np.zeros((n, n, 4), np.uint16), np.zeros((n, n, 8), np.float32)
        assert frame_fn(n, h0a, om, C.byref(cf), T, disp, norm, foam, f32) == 0
        ref = g.f32(0)
        for c, name in enumerate(H.CHANNELS):
            if name == "foam":
                assert np.abs(f32[..., c] - ref[..., c]).max() <= H.TOL_FOAM_ABS
            else:
                assert H.relmax(f32[..., c], ref[..., c]) < 5e-6, (frame, name)
        assert H.fp16_close(disp, g.displacement(0)) <= 1.0
        assert H.fp16_close(norm[..., :3], g.normal(0)[..., :3]) <= 1.0


@pytest.mark.parametrize("n", [256, 512, 1024])
@pytest.mark.parametrize("slots", [1, 3, 4, 8])
def test_tick_group_items_cover_every_row_group_exactly_once(emul, n, slots):
    """ow_run's tick groups (k_tick_group_c_lp): the pass-1 item decode (TickPlan::decode, compiled here as plain C++) hands out, for
    every cascade, layer 0 and layer 2 of every 8-row group, layer 1 of the upper-half groups only, and the three row-0 transforms --
    each exactly once; pass 2 has one item per plan_lp_rows rows."""
    out = np.zeros((slots * (n // 8) * 4 + 64, 3), np.int32)
    emul.emul_tick_items.argtypes = [C.c_int, C.c_int, np.ctypeslib.ndpointer(np.int32, flags="C")]
    cnt = emul.emul_tick_items(n, slots, out)
    got = sorted(map(tuple, out[:cnt].tolist()))
    want = []
    for s_ in range(slots):
        for g in range(n // 8):
            want += [(0, s_, 8 * g), (2, s_, 8 * g)]
            if 8 * g >= n // 2:
                want.append((1, s_, 8 * g))
        want += [(3, s_, 0), (4, s_, 0), (5, s_, 0)]
    assert got == sorted(want)
    rows_per_item = max(1, 128 // (n // 16))
    assert emul.emul_tick_items_2(n, slots) == slots * n // rows_per_item
    # the k_pass1c-shaped items: every 8-row group exactly once, and the groups that share a block lie in the same half of the rows
    # (the block barriers of the layer loop, and "skip layer 1 below N/2", are block-uniform)
    emul.emul_tick_items_compact.argtypes = emul.emul_tick_items.argtypes
    cnt = emul.emul_tick_items_compact(n, slots, out)
    items = out[:cnt]
    assert sorted(map(tuple, items[:, 1:].tolist())) == sorted((s_, 8 * g) for s_ in range(slots) for g in range(n // 8))
    for block in np.unique(items[:, 0]):
        rows = items[items[:, 0] == block]
        assert len(set(rows[:, 1].tolist())) == 1 and len(set((rows[:, 2] >= n // 2).tolist())) == 1
