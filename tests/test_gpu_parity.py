"""Parity tests proper: the HIP path (through the C-ABI / WaveGenerator mirror) against the CPU oracle on
the same seeded inputs.  Tolerances are the ones of north_star / SURVEY.md 8c and are written where used."""
import numpy as np
import pytest

import helpers as H
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, _lib
from godotoceanwaves_amd.presets import DEPTH, UPDATE_DELTA, cascade_preset
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def make_gen(n, cascade_ids, debug=True, kernels=None):
    gen = WaveGenerator()
    gen.map_size = n
    gen.debug_f32 = debug
    gen.kernels = kernels
    gen.init_gpu(max(2, len(cascade_ids)))
    params = [WaveCascadeParameters(**cascade_preset(ci)) for ci in cascade_ids]
    return gen, params


@pytest.mark.parametrize("n", [128, 256, 512, 1024])
def test_spectrum_and_omega(n):
    """spectrum_compute.glsl: h0 within 2e-5 (max-norm relative; libm vs device libm ulps), omega BIT-exact."""
    ids = [0, 2]
    gen, params = make_gen(n, ids)
    gen.update_all(UPDATE_DELTA, params)
    gen.sync()
    for i, ci in enumerate(ids):
        p = cascade_preset(ci)
        h0, om = gen.get_spectrum(i)
        ref = O.spectrum_compute(n, H.spectrum_pc(p))
        assert np.isfinite(h0).all()
        assert H.relmax(h0, ref) < 2e-5
        om_ref = O.omega(n, p["tile_length"], DEPTH)
        mism = int((om.view(np.uint32) != om_ref.view(np.uint32)).sum())
        assert mism <= 2, f"{mism} omega texels differ"  # P(double-rounding disagreement) ~ 2^-28 per texel


@pytest.mark.parametrize("kernels", ["standard", "layer_parallel", "layer_parallel_compact"])
@pytest.mark.parametrize("n,ids", [(128, [0]), (256, [0, 1, 2, 3]), (512, [2, 4]), (1024, [2])])
def test_frame_parity_vs_oracle(n, ids, kernels):
    """3 frames of modulate + IFFT + unpack: FP32 channels <= 1e-4 max-norm relative, FP16 maps <= 1 ulp,
    intermediate (after the first row pass + transpose) <= 1e-5."""
    gen, params = make_gen(n, ids, kernels=kernels)
    og = H.oracle_generator(n, ids)
    for frame in range(3):
        gen.update_all(UPDATE_DELTA, params)
        og.update_all(UPDATE_DELTA)
        gen.sync()
        for i in range(len(ids)):
            assert params[i].time == og.params[i].time
            f32, ref = gen.get_maps_f32(i), og.f32(i)
            for c, name in enumerate(H.CHANNELS):
                if name == "foam":
                    assert np.abs(f32[..., c] - ref[..., c]).max() <= H.TOL_FOAM_ABS, (frame, i, name)
                else:
                    assert H.relmax(f32[..., c], ref[..., c]) < H.TOL_F32, (frame, i, name)
            disp, norm = gen.get_maps(i)
            assert H.quantisation_exact(f32, disp, norm)  # the maps ARE the RTE quantisation of the FP32 channels
            assert H.fp16_close(disp, og.displacement(i)) <= 1.0
            assert H.fp16_close(norm[..., :3], og.normal(i)[..., :3]) <= 1.0
            assert np.abs(norm[..., 3].astype(np.float64) - og.normal(i)[..., 3].view(np.float16).astype(np.float64)).max() <= H.TOL_FOAM_ABS
    if gen.last_kernel_family() == "layer_parallel_compact":   # (n = 128 falls back to the four-layer kernels)
        with pytest.raises(_lib.OceanWavesError):
            gen.get_intermediate(len(ids) - 1)
        return
    # the fft_buffer contents after pass 1 (reference: fft_compute + transpose, half 0)
    inter = gen.get_intermediate(len(ids) - 1)
    p = cascade_preset(ids[-1])
    spec = og.spectrum(len(ids) - 1)
    x = O.spectrum_modulate(n, p["tile_length"], DEPTH, np.float32(og.params[len(ids) - 1].time), spec)
    tab = O.fft_butterfly(n)
    ref_half0 = O.fft_rows(n, tab, x).transpose(0, 2, 1, 3)
    assert H.relmax(inter, ref_half0) < 1e-5


def test_process_one_cascade_per_frame_equals_update_all():
    """wave_generator.gd:56-63,90-109: update() arms, _process() drains highest index first, the next update()
    flushes leftovers 0..remaining-1 -- results identical (bitwise) to the batched launch."""
    n, ids = 256, [0, 1, 2]
    gen_a, pa = make_gen(n, ids, debug=False)
    gen_b, pb = make_gen(n, ids, debug=False)
    for tick in range(3):
        gen_a.update_all(UPDATE_DELTA, pa)
        gen_b.update(UPDATE_DELTA, pb)
        assert gen_b.pass_num_cascades_remaining == 3
        if tick != 1:
            for k in range(3):
                gen_b._process(0.0)
                assert gen_b.pass_num_cascades_remaining == 2 - k
        else:
            gen_b._process(0.0)  # only cascade 2; cascades 0,1 are flushed by the next update()
            assert gen_b.pass_num_cascades_remaining == 2
    gen_b.update(UPDATE_DELTA, pb)   # flush
    gen_a.update_all(UPDATE_DELTA, pa)
    for k in range(3):
        gen_b._process(0.0)
    gen_a.sync(); gen_b.sync()
    for i in range(3):
        da, na = gen_a.get_maps(i)
        db, nb = gen_b.get_maps(i)
        assert np.array_equal(da.view(np.uint16), db.view(np.uint16))
        assert np.array_equal(na.view(np.uint16), nb.view(np.uint16))
        assert pa[i].time == pb[i].time and not pb[i].should_generate_spectrum


def test_dirty_flag_regenerates_spectrum():
    n = 256
    gen, params = make_gen(n, [0, 1], debug=False)
    gen.update_all(UPDATE_DELTA, params); gen.sync()
    h0_before, _ = gen.get_spectrum(0)
    params[0].wind_speed = 14.0            # setter sets should_generate_spectrum (wave_cascade_parameters.gd:15)
    assert params[0].should_generate_spectrum
    gen.update_all(UPDATE_DELTA, params); gen.sync()
    h0_after, _ = gen.get_spectrum(0)
    assert not params[0].should_generate_spectrum and not np.array_equal(h0_before, h0_after)
    p = cascade_preset(0); p["wind_speed"] = 14.0
    assert H.relmax(h0_after, O.spectrum_compute(n, H.spectrum_pc(p))) < 2e-5


def test_foam_state_roundtrip():
    """the only persistent state besides `time` is normal.a (FP16): save/restore reproduces the trajectory bitwise"""
    n = 256
    gen, params = make_gen(n, [0, 2], debug=False)
    for _ in range(4):
        gen.update_all(UPDATE_DELTA, params)
    gen.sync()
    saved = [gen.get_maps(i)[1].copy() for i in range(2)]
    times = [p.time for p in params]
    for _ in range(3):
        gen.update_all(UPDATE_DELTA, params)
    gen.sync()
    final = [gen.get_maps(i)[1].copy() for i in range(2)]
    assert float(final[0][..., 3].max()) > 0.0  # foam exists for cascade 0 (foam_amount 8)
    gen2, params2 = make_gen(n, [0, 2], debug=False)
    for i in range(2):
        params2[i].time = times[i]
    gen2.update_all(0.0, params2)          # generates the spectra; one throw-away frame
    for i in range(2):
        gen2.set_normal_map(i, saved[i])
    for _ in range(3):
        gen2.update_all(UPDATE_DELTA, params2)
    gen2.sync()
    for i in range(2):
        assert np.array_equal(gen2.get_maps(i)[1].view(np.uint16), final[i].view(np.uint16))


@pytest.mark.parametrize("n", [1024, 2048])
def test_full_size_properties(n):
    """BASELINE sizes, size-independent checks: (a) against NumPy's FP64 ifft2 fed with the device's own
    h0/omega (identity: result == (N^2 ifft2 X)^T) within 1e-4; (b) displacement mean == DC term."""
    import np_twin as T
    ids = [1]
    gen, params = make_gen(n, ids)
    gen.update_all(UPDATE_DELTA, params); gen.sync()
    h0, om = gen.get_spectrum(0)
    p = cascade_preset(ids[0])
    t32 = np.float32(params[0].time)
    phase32 = om * t32                                   # the FP32-rounded product of spectrum_modulate.glsl:65
    x = T.modulate(n, p["tile_length"], DEPTH, float(t32),
                   (h0[..., 0] + 1j * h0[..., 1]).astype(np.complex128), (h0[..., 2] + 1j * h0[..., 3]).astype(np.complex128),
                   omega=phase32.astype(np.float64) / float(t32))
    ref = T.unpack(T.ifft2_ref(x), p["whitecap"], params[0].foam_grow_rate, params[0].foam_decay_rate)
    f32 = gen.get_maps_f32(0)
    for c, name in enumerate(H.CHANNELS):
        if name == "foam":
            assert np.abs(f32[..., c] - ref[..., c]).max() <= H.TOL_FOAM_ABS
        else:
            assert H.relmax(f32[..., c], ref[..., c]) < H.TOL_F32, name


@pytest.mark.parametrize("kernels", [None, "standard"])
@pytest.mark.parametrize("n,ids", [(1024, [0, 1, 2, 3, 4]), (1024, [0, 1, 2, 3, 4, 5, 6]), (1024, [0, 1, 2, 3, 4, 5, 6, 7]), (2048, [0, 2]), (2048, [0, 1, 2])])
def test_many_cascades_and_batched_launches_match_oracle(n, ids, kernels):
    """Five cascades of 1024^2 in one pair of launches, seven and MAX_CASCADES = 8 (water.gdshader:8; BASELINE config C4's per-node
    total) as 4 + 3 and 4 + 4, and more cascades than one pair takes at 2048^2 (the runtime batches
    there at 4 Mi texels = one cascade and reuses the scratch intermediate between batches): every cascade still matches the
    oracle, two frames.  kernels=None is the runtime's own choice (the compact-intermediate kernels at these sizes)."""
    gen, params = make_gen(n, ids, kernels=kernels)
    og = H.oracle_generator(n, ids)
    for frame in range(2):
        gen.update_all(UPDATE_DELTA, params)
        og.update_all(UPDATE_DELTA)
    gen.sync()
    for i in range(len(ids)):
        f32, ref = gen.get_maps_f32(i), og.f32(i)
        for c, name in enumerate(H.CHANNELS):
            if name == "foam":
                assert np.abs(f32[..., c] - ref[..., c]).max() <= H.TOL_FOAM_ABS, (i, name)
            else:
                assert H.relmax(f32[..., c], ref[..., c]) < H.TOL_F32, (i, name)
    family = gen.last_kernel_family()        # of the LAST batch
    assert family == ("standard" if kernels == "standard" else "compact")
    split = n == 2048 or len(ids) > 6        # 1024^2: one pair up to 6 cascades, 7 go as 4 + 3
    assert gen.last_batch_cascades() == (1 if n == 2048 else {7: 3, 8: 4}.get(len(ids), len(ids)))
    if split:       # drained highest index first: only the last batch's (lowest indices') intermediate is still there
        with pytest.raises(_lib.OceanWavesError):
            gen.get_intermediate(len(ids) - 1)
    if family == "compact":
        with pytest.raises(_lib.OceanWavesError):
            gen.get_intermediate(0)          # the compact intermediate has no counterpart in the reference's fft_buffer
    else:
        assert gen.get_intermediate(0).shape == (4, n, n, 2)
        if not split:
            assert gen.get_intermediate(len(ids) - 1).shape == (4, n, n, 2)   # one batch: every cascade's intermediate is there


@pytest.mark.parametrize("n,count", [(1024, 4), (256, 4), (2048, 4), (2048, 8), (1024, 8)],
                         ids=["C3_headline_1024x4", "C2_256x4", "C5_2048x4", "2048x8", "C4_total_1024x8"])
def test_baseline_configs_through_ow_run_match_the_oracle(n, count):
    """The exact BASELINE configurations -- C3 = 1024^2 x 4 (the headline: the seamless stream of tick pairs), C2 = 256^2 x 4 (tick groups),
    C5 = 2048^2 x 4 (LDS-tiling stress), 2048^2 x 8, and 1024^2 x 8 (C4's per-node total) -- through ow_run, the call bench.py times, in
    the merged launches each size has (tick pairs / tick groups).  Every FP32 channel <= 1e-4 of the oracle (max-norm relative), the RGBA16F maps within one FP16 ulp (+ 1e-5 of the channel
    maximum) of the oracle's and exactly the RTE quantisation of the FP32 channels, foam within one FP16 step; three ticks."""
    ids = list(range(count))
    gen, params = make_gen(n, ids)
    og = H.oracle_generator(n, ids)
    gen.run(UPDATE_DELTA, params, 3)
    for _ in range(3):
        og.update_all(UPDATE_DELTA)
    gen.sync()
    assert gen.last_kernel_family() in (("tick_groups_compact",) if n == 256 else ("compact", "tick_pairs_compact"))
    worst = 0.0
    for i in range(count):
        assert params[i].time == og.params[i].time
        f32, ref = gen.get_maps_f32(i), og.f32(i)
        for c, name in enumerate(H.CHANNELS):
            if name == "foam":
                assert np.abs(f32[..., c] - ref[..., c]).max() <= H.TOL_FOAM_ABS, (i, name)
            else:
                e = H.relmax(f32[..., c], ref[..., c])
                worst = max(worst, e)
                assert e < H.TOL_F32, (i, name, e)
        disp, norm = gen.get_maps(i)
        assert H.quantisation_exact(f32, disp, norm)
        assert H.fp16_close(disp, og.displacement(i)) <= 1.0, i
        assert H.fp16_close(norm[..., :3], og.normal(i)[..., :3]) <= 1.0, i
        assert np.abs(norm[..., 3].astype(np.float64) - og.normal(i)[..., 3].view(np.float16).astype(np.float64)).max() <= H.TOL_FOAM_ABS
    print(f"{n}^2 x {count} through ow_run: worst FP32 channel error {worst:.2e}")


def _edge_cases():
    from edge_presets import edge_presets
    return sorted(edge_presets().items())


@pytest.mark.parametrize("name,preset", _edge_cases(), ids=[k for k, _ in _edge_cases()])
def test_parameter_range_edges_match_oracle(name, preset):
    """range ends of every exported parameter, non-square tiles, wrapping seeds, t = 0 and the largest phases (256^2)"""
    n = 256
    gen = WaveGenerator()
    gen.map_size, gen.debug_f32 = n, True
    gen.init_gpu(2)
    params = [WaveCascadeParameters(**preset)]
    og = O.Generator(n, 1, DEPTH)
    H.set_params(og.params[0], preset)
    for _ in range(2):
        gen.update_all(UPDATE_DELTA, params)
        og.update_all(UPDATE_DELTA)
    gen.sync()
    f32, ref = gen.get_maps_f32(0), og.f32(0)
    assert np.isfinite(f32).all()
    for c, cname in enumerate(H.CHANNELS):
        if cname == "foam":
            assert np.abs(f32[..., c] - ref[..., c]).max() <= H.TOL_FOAM_ABS, cname
        elif np.abs(ref[..., c]).max() > 0:
            assert H.relmax(f32[..., c], ref[..., c]) < H.TOL_F32, cname
    disp, norm = gen.get_maps(0)
    assert H.quantisation_exact(f32, disp, norm)


def test_invalid_arguments_are_errors():
    gen, params = make_gen(256, [0, 1], debug=False)
    L = _lib.load()
    with pytest.raises(_lib.OceanWavesError):
        gen.update_all(UPDATE_DELTA, params + [WaveCascadeParameters()])  # more cascades than allocated
    with pytest.raises(_lib.OceanWavesError):
        gen.get_maps_f32(0)                                             # context created without DEBUG_F32
    with pytest.raises(_lib.OceanWavesError):
        gen.get_maps(5)


@pytest.mark.parametrize("n,ids", [(1024, [0, 2]), (1024, [1, 3, 4, 5, 6]), (2048, [2]), (512, [0, 1, 2, 3, 4, 5]), (256, [0, 7])])
def test_compact_intermediate_kernels_match_oracle(n, ids):
    """The three-layer intermediate (Pass1::layer_input_c, tests/test_compact_math.py) against the oracle: same tolerances as
    the reference-layout kernels, including the Nyquist row / column the closed forms reproduce; the debug view of the
    intermediate is refused for such batches."""
    gen, params = make_gen(n, ids, kernels="compact")
    og = H.oracle_generator(n, ids)
    for frame in range(3):
        gen.update_all(UPDATE_DELTA, params)
        og.update_all(UPDATE_DELTA)
    gen.sync()
    for i in range(len(ids)):
        f32, ref = gen.get_maps_f32(i), og.f32(i)
        for c, name in enumerate(H.CHANNELS):
            if name == "foam":
                assert np.abs(f32[..., c] - ref[..., c]).max() <= H.TOL_FOAM_ABS, (i, name)
            else:
                assert H.relmax(f32[..., c], ref[..., c]) < 1e-5, (i, name, H.relmax(f32[..., c], ref[..., c]))
                # the two Nyquist lines on their own (row 0 and column 0 of the map carry no special role in the OUTPUT;
                # the lines live in the spectrum -- so compare the worst texel row / column as well)
        disp, norm = gen.get_maps(i)
        assert H.quantisation_exact(f32, disp, norm)
        assert H.fp16_close(disp, og.displacement(i)) <= 1.0
        assert H.fp16_close(norm[..., :3], og.normal(i)[..., :3]) <= 1.0
    with pytest.raises(_lib.OceanWavesError) as e:
        gen.get_intermediate(0)
    assert e.value.status == _lib.OW_ERR_STATE


@pytest.mark.parametrize("names", [("non_square_tile", "late_time"), ("gale_long_fetch", "whitecap_foam_extremes"),
                                   ("wrapping_seed", "swell_spread_detail_extremes")])
def test_compact_kernels_on_parameter_range_edges(names):
    """the closed-form Nyquist-line handling of the compact intermediate on non-square tiles, t = 0, the largest phases,
    the smallest and largest tiles (1024^2, two cascades per launch so that the runtime picks the compact kernels)"""
    from edge_presets import edge_presets
    n, presets = 1024, [edge_presets()[k] for k in names]
    gen = WaveGenerator()
    gen.map_size, gen.debug_f32 = n, True
    gen.init_gpu(2)
    params = [WaveCascadeParameters(**p) for p in presets]
    og = O.Generator(n, 2, DEPTH)
    for i, p in enumerate(presets):
        H.set_params(og.params[i], p)
    for _ in range(2):
        gen.update_all(UPDATE_DELTA, params)
        og.update_all(UPDATE_DELTA)
    gen.sync()
    assert gen.last_kernel_family() == "compact"
    for i in range(2):
        f32, ref = gen.get_maps_f32(i), og.f32(i)
        assert np.isfinite(f32).all()
        for c, cname in enumerate(H.CHANNELS):
            if cname == "foam":
                assert np.abs(f32[..., c] - ref[..., c]).max() <= H.TOL_FOAM_ABS, (names[i], cname)
            elif np.abs(ref[..., c]).max() > 0:
                assert H.relmax(f32[..., c], ref[..., c]) < H.TOL_F32, (names[i], cname, H.relmax(f32[..., c], ref[..., c]))


@pytest.mark.parametrize("names", [("non_square_tile", "late_time"), ("wrapping_seed", "whitecap_foam_extremes")])
def test_split_plan_pass1_on_parameter_range_edges(names):
    """2048^2 (beyond the reference's sizes, BASELINE config C5): pass 1 is the split plan there (k_pass1c_split: one wave per parity
    of the element index, the radix-2 join done by the storing threads, texel row 0 by its own wave pair).  Non-square tiles, t = 0
    with a wrapping seed, the largest phases and the smallest tile, two cascades = two batches."""
    from edge_presets import edge_presets
    n, presets = 2048, [edge_presets()[k] for k in names]
    gen = WaveGenerator()
    gen.map_size, gen.debug_f32 = n, True
    gen.init_gpu(2)
    params = [WaveCascadeParameters(**p) for p in presets]
    og = O.Generator(n, 2, DEPTH)
    for i, p in enumerate(presets):
        H.set_params(og.params[i], p)
    for _ in range(2):
        gen.update_all(UPDATE_DELTA, params)
        og.update_all(UPDATE_DELTA)
    gen.sync()
    assert gen.last_kernel_family() == "compact" and gen.last_batch_cascades() == 1
    for i in range(2):
        f32, ref = gen.get_maps_f32(i), og.f32(i)
        assert np.isfinite(f32).all()
        for c, cname in enumerate(H.CHANNELS):
            if cname == "foam":
                assert np.abs(f32[..., c] - ref[..., c]).max() <= H.TOL_FOAM_ABS, (i, cname)
            elif np.abs(ref[..., c]).max() > 0:
                assert H.relmax(f32[..., c], ref[..., c]) < H.TOL_F32, (i, cname)
        disp, norm = gen.get_maps(i)
        assert H.quantisation_exact(f32, disp, norm)


@pytest.mark.parametrize("n,ids,family", [(1024, [0, 1, 2], "compact"), (512, [0, 1, 2, 3, 4, 5, 6, 7], "compact"),
                                          (256, [0, 1, 2, 3, 4, 5, 6, 7], "layer_parallel_compact"), (128, [0, 1, 2, 3, 4, 5, 6, 7], "layer_parallel"),
                                          (2048, [1], "compact")])
def test_runtime_kernel_choice_and_parity(n, ids, family):
    """what the runtime picks on its own for odd cascade counts and sizes, and that each choice matches the oracle (two frames)"""
    gen, params = make_gen(n, ids)
    og = H.oracle_generator(n, ids)
    for _ in range(2):
        gen.update_all(UPDATE_DELTA, params)
        og.update_all(UPDATE_DELTA)
    gen.sync()
    assert gen.last_kernel_family() == family
    for i in range(len(ids)):
        f32, ref = gen.get_maps_f32(i), og.f32(i)
        for c, name in enumerate(H.CHANNELS):
            if name == "foam":
                assert np.abs(f32[..., c] - ref[..., c]).max() <= H.TOL_FOAM_ABS, (i, name)
            else:
                assert H.relmax(f32[..., c], ref[..., c]) < H.TOL_F32, (i, name)
        disp, norm = gen.get_maps(i)
        assert H.quantisation_exact(f32, disp, norm)


def test_live_parameter_edits_between_ticks_compact_family():
    """the reference's UI edits parameters while the simulation runs (wave_cascade_parameters.gd setters raise the dirty flag):
    a spectrum regeneration and a non-spectrum edit between ticks, on the runtime's default kernels at 1024^2 x 2, with a
    non-default depth"""
    n, ids = 1024, [0, 2]
    gen = WaveGenerator()
    gen.map_size, gen.debug_f32, gen.depth = n, True, 35.0
    gen.init_gpu(2)
    params = [WaveCascadeParameters(**cascade_preset(ci)) for ci in ids]
    og = O.Generator(n, 2, 35.0)
    for i, ci in enumerate(ids):
        H.set_params(og.params[i], cascade_preset(ci))
    for frame in range(4):
        if frame == 2:
            params[0].wind_speed = 14.0          # regenerates cascade 0's spectrum on the next update
            params[1].tile_length = (21.0, 34.0)
            og.params[0].wind_speed = 14.0
            og.params[0].should_generate_spectrum = 1
            og.params[1].tile_length[0], og.params[1].tile_length[1] = 21.0, 34.0
            og.params[1].should_generate_spectrum = 1
        if frame == 3:
            params[1].foam_amount = 2.5           # the reference regenerates on this one too (wave_cascade_parameters.gd:32-35)
            og.params[1].foam_amount = 2.5
            og.params[1].should_generate_spectrum = 1
        gen.update_all(UPDATE_DELTA, params)
        og.update_all(UPDATE_DELTA)
    gen.sync()
    assert gen.last_kernel_family() == "compact"
    for i in range(2):
        f32, ref = gen.get_maps_f32(i), og.f32(i)
        for c, name in enumerate(H.CHANNELS):
            if name == "foam":
                assert np.abs(f32[..., c] - ref[..., c]).max() <= H.TOL_FOAM_ABS, (i, name)
            else:
                assert H.relmax(f32[..., c], ref[..., c]) < H.TOL_F32, (i, name, H.relmax(f32[..., c], ref[..., c]))


def test_long_run_matches_the_oracle_at_the_final_time():
    """20 000 ticks of the headline configuration (400 s of simulated time, phases up to ~3e4 rad).  Everything but foam is a
    function of the time alone, so the oracle can be started one tick before the end: the HIP path, after 20 010 ticks of
    FP64 `time += delta` on the host (wave_generator.gd:103) and FP32 phases on the device, must still agree with it; foam
    (recurrent) stays in [0, 1]."""
    n, ids, ticks = 1024, [0, 1, 2, 3], 20000
    gen, params = make_gen(n, ids)
    gen.run(UPDATE_DELTA, params, 10)
    t_expect = [p.time for p in params]
    gen.run(UPDATE_DELTA, params, ticks)
    gen.sync()
    og = H.oracle_generator(n, ids)
    for i in range(4):
        for _ in range(ticks):
            t_expect[i] += UPDATE_DELTA
        assert params[i].time == t_expect[i]
    # one oracle tick ending at the same time: rewind by the delta it is about to add
    for i in range(4):
        og.params[i].time = t_expect[i] - UPDATE_DELTA
    og.update_all(UPDATE_DELTA)
    for i in range(4):
        f32, ref = gen.get_maps_f32(i), og.f32(i)
        assert np.isfinite(f32).all()
        for c, name in enumerate(H.CHANNELS):
            if name == "foam":
                assert f32[..., c].min() >= 0.0 and f32[..., c].max() <= 1.0
            else:
                # time - delta + delta is not bit for bit the accumulated time: allow the phase of one FP64 ulp at 520 s (nothing)
                # plus the usual tolerance
                assert H.relmax(f32[..., c], ref[..., c]) < H.TOL_F32, (i, name, H.relmax(f32[..., c], ref[..., c]))


@pytest.mark.parametrize("n,ci,t0", [(1024, 2, 86400.0), (256, 7, 14400.0), (1024, 0, 3600.0)])
def test_parity_holds_at_the_phases_of_a_long_session(n, ci, t0):
    """omega * t reaches 1e5 .. 1e6 rad after hours of simulated time -- beyond the range the three-step Cody-Waite reduction of
    sincos_phase is exact for (ow_device.h).  What matters is the result: the energy sits at low omega, and the FP32 channels stay
    within 1e-5 of the oracle (glibc sinf / cosf of the same FP32 phase) after a simulated day (profiles/r03_long_time_parity.txt).
    (The reference itself degrades earlier: `time` is an FP32 push constant, 8 ms per ulp at 86 400 s.)"""
    gen = WaveGenerator()
    gen.map_size = n
    gen.debug_f32 = True
    gen.init_gpu(2)
    params = [WaveCascadeParameters(**cascade_preset(ci))]
    params[0].time = t0
    og = H.oracle_generator(n, [ci])
    og.params[0].time = t0
    for _ in range(2):
        gen.update_all(UPDATE_DELTA, params)
        og.update_all(UPDATE_DELTA)
    gen.sync()
    f32, ref = gen.get_maps_f32(0), og.f32(0)
    for c, name in enumerate(H.CHANNELS):
        if name != "foam":
            assert H.relmax(f32[..., c], ref[..., c]) < 2e-5, name
    assert H.fp16_close(gen.get_maps(0)[0], og.displacement(0)) <= 1.0
