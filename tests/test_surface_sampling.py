"""Consumer-side sampling: the sums water.gdshader and sea_spray_particle.gdshader take over the
two array textures, and the sea-spray spawn mask.  CPU: the oracle's restatement of those shaders against the
independent NumPy FP64 view (tests/consumer.py) and against hand-built cases.  GPU: ow_sample_surface against the
oracle on the maps the HIP path produced -- same FP32 operations in the same order, so the comparison is bit for bit
(both sides are built with FP contraction off; division and square root are correctly rounded on either)."""
import numpy as np
import pytest

import consumer as K
import helpers as H
from godotoceanwaves_amd import _lib
from godotoceanwaves_amd.presets import UPDATE_DELTA, cascade_preset
from oracle import oracle as O
from oracle import ref as R

SCALES3 = np.array([[1 / 88, 1 / 88, 1.0, 1.0], [1 / 57, 1 / 57, 0.75, 0.5], [1 / 16, 1 / 16, 0.5, 0.25]], np.float32)


def random_maps(C, N, seed=0, foam_hi=0.6):
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((C, N, N, 4)).astype(np.float16)
    m = (rng.standard_normal((C, N, N, 4)) * 0.3).astype(np.float16)
    m[..., 3] = rng.uniform(0, foam_hi, (C, N, N)).astype(np.float16)
    return d, m


def query_points(count, seed=1, span=700.0):
    """random points plus the awkward ones: origin, texel centres and edges, negative coordinates, far away"""
    rng = np.random.default_rng(seed)
    xz = rng.uniform(-span, span, (count, 2)).astype(np.float32)
    xz[:8] = [[0, 0], [88 / 256 * 0.5, 88 / 256 * 0.5], [88.0, 88.0], [-88.0, 57.0], [-0.001, -0.001], [1e4, -1e4], [16.0, -16.0], [44.0, 28.5]]
    return xz


def test_record_layout_is_the_same_on_every_side():
    from godotoceanwaves_amd import WaveGenerator
    assert O.SURFACE_SAMPLE.itemsize == 64 and WaveGenerator.SURFACE_SAMPLE == O.SURFACE_SAMPLE


FIELDS_PINNED = ["displacement", "gradient", "foam", "normal_factor", "foam_factor", "scale_factor", "spray_active",
                 "gradient_fragment", "foam_fragment"]


@pytest.mark.skipif(not R.available(), reason="oracle/_ref/libglsl_ref.so not built (needs the reference checkout)")
@pytest.mark.parametrize("case", ["random_maps", "generated_maps", "spray_active", "fine_cascades"])
def test_oracle_sampling_is_bit_exact_against_the_reference_shader_text(case):
    """owo_sample_surface against the reference's OWN statements: cubic_weights / texture_bicubic (water.gdshader:41-68), the
    cascade loops of vertex() (:31-37) and fragment() (:72-82), the particle shader's spawn decision
    (sea_spray_particle.gdshader:80-89) and displacement sum (:103-107), compiled from the .gdshader files through
    oracle/glsl_shim.h (texture() = GL_LINEAR + GL_REPEAT with exact weights, defined there)."""
    sc = SCALES3
    if case == "random_maps":
        d, m = random_maps(3, 64)
    elif case == "spray_active":  # foam near 1 and flat normals: ACTIVE both ways, both factor branches
        d, m = random_maps(3, 32, seed=5, foam_hi=0.7)
        m[..., :2] *= np.float16(0.1)
    elif case == "fine_cascades":  # ppm * 0.1 >= 1 on every cascade: the mix takes the bilinear lookup alone
        d, m = random_maps(3, 64, seed=7)
        sc = np.array([[1 / 5, 1 / 4, 1.0, 1.3], [1 / 3, 1 / 6, 0.75, 0.5], [1 / 2, 1 / 2, 0.5, 0.25]], np.float32)
    else:  # maps the pipeline itself produced (oracle generator), three cascades
        g = H.oracle_generator(128, [0, 1, 2])
        for _ in range(2):
            g.update_all(UPDATE_DELTA)
        d = np.stack([g.displacement(i) for i in range(3)])
        m = np.stack([g.normal(i) for i in range(3)])
    xz = query_points(3000, seed=11)
    o = O.sample_surface(d, m, sc, xz)
    r, dp = R.sample_surface(d, m, sc, xz)
    for f in FIELDS_PINNED:
        assert np.array_equal(o[f].view(np.uint32), r[f].view(np.uint32)), f
    # the particle shader's own displacement sum (:103-107) is the vertex shader's (:31-37)
    assert np.array_equal(dp.view(np.uint32), o["displacement"].view(np.uint32))
    if case == "spray_active":
        assert 0 < o["spray_active"].sum() < len(xz)
    if case == "fine_cascades":  # gradient_scaled = the bilinear operand of water.gdshader:81's mix: pinned where the mix factor is 1
        assert np.array_equal(o["gradient_scaled"].view(np.uint32), r["gradient_fragment"].view(np.uint32))
        assert np.array_equal(o["foam"].view(np.uint32), r["foam_fragment"].view(np.uint32))


def test_oracle_sampling_agrees_with_fp64_consumer_view():
    d, m = random_maps(3, 64)
    xz = query_points(2000)
    o = O.sample_surface(d, m, SCALES3, xz)
    x, z, sc = xz[:, 0].astype(np.float64), xz[:, 1].astype(np.float64), SCALES3.astype(np.float64)
    ref = K.displacement_at([d[i] for i in range(3)], sc, x, z)
    # FP32 texture coordinates far from the origin carry ~1e-4 texel of rounding
    assert np.abs(o["displacement"] - ref).max() < 2e-3
    near = np.abs(xz).max(axis=1) < 100.0
    assert np.abs(o["displacement"] - ref)[near].max() < 2e-4
    g = sum(K.texture_bilinear(m[i].astype(np.float64), x * sc[i, 0], z * sc[i, 1]) for i in range(3))
    gs = sum(K.texture_bilinear(m[i].astype(np.float64), x * sc[i, 0], z * sc[i, 1])[..., :2] * sc[i, 3] for i in range(3))
    assert np.abs(o["gradient"] - g[..., :2])[near].max() < 1e-4 and np.abs(o["foam"] - g[..., 3])[near].max() < 1e-4
    assert np.abs(o["gradient_scaled"] - gs)[near].max() < 1e-4


def test_oracle_fragment_filter_agrees_with_the_direct_b_spline_sum():
    """water.gdshader:41-82: the four-bilinear-tap cubic B-spline and its mix with the bilinear lookup, against the direct
    16-tap sum; one cascade coarse enough (and one fine enough) for the mix factor to be below / at 1"""
    d, m = random_maps(3, 64)
    sc = np.array([[1 / 880, 1 / 880, 1.0, 1.3], [1 / 5, 1 / 4, 0.75, 0.5], [1 / 16, 1 / 9, 0.5, 0.25]], np.float32)
    assert min(1.0, 0.1 * 64 * sc[0, 0]) < 1 and min(1.0, 0.1 * 64 * min(sc[1, 0], sc[1, 1])) == 1.0
    xz = query_points(2000, span=300.0)
    o = O.sample_surface(d, m, sc, xz)
    ref = K.gradient_fragment_at([m[i] for i in range(3)], sc.astype(np.float64), xz[:, 0].astype(np.float64), xz[:, 1].astype(np.float64))
    near = np.abs(xz).max(axis=1) < 100.0
    assert np.abs(o["gradient_fragment"] - ref[:, :2])[near].max() < 2e-4
    assert np.abs(o["foam_fragment"] - ref[:, 2])[near].max() < 2e-4
    assert np.abs(o["gradient_fragment"] - o["gradient_scaled"]).max() > 1e-2     # the filter is not a no-op


def test_spray_mask_cases():
    """sea_spray_particle.gdshader:83-89 on constant maps: flat + full foam spawns at full size; little foam never
    spawns; a steep surface (normal.y below the window's lower extrapolation) does not spawn either"""
    N, sc = 16, np.array([[1 / 50, 1 / 50, 1.0, 1.0]], np.float32)
    xz = np.array([[3.0, 7.0], [-11.0, 120.0]], np.float32)

    def run(gx, gy, foam):
        d = np.zeros((1, N, N, 4), np.float16)
        m = np.zeros((1, N, N, 4), np.float16)
        m[..., 0], m[..., 1], m[..., 3] = gx, gy, foam
        return O.sample_surface(d, m, sc, xz)

    flat = run(0.0, 0.0, 1.0)
    assert flat["spray_active"].all() and np.allclose(flat["normal_factor"], 1.0) and np.allclose(flat["foam_factor"], 1.0)
    assert np.allclose(flat["scale_factor"], 1.0)
    assert not run(0.0, 0.0, 0.5)["spray_active"].any()
    assert not run(0.0, 0.0, 0.9)["spray_active"].any()          # foam > 0.9 is strict
    half = run(0.0, 0.0, 0.95)                                   # middle of the foam window
    assert half["spray_active"].all() and np.allclose(half["foam_factor"], 0.625, atol=2e-3)
    steep = run(1.0, 1.0, 1.0)                                   # normal.y = 1/sqrt(3) = 0.577: factor = 0.25 + 0.75*(-4.9) < 0
    assert not steep["spray_active"].any() and (steep["normal_factor"] < 0).all()
    slope = run(0.3, 0.0, 1.0)                                   # normal.y = 0.958: inside the window
    ny = 1.0 / np.sqrt(1.09)
    assert slope["spray_active"].all() and np.allclose(slope["normal_factor"], 0.25 + 0.75 * (ny - 0.92) / 0.07, atol=1e-3)


def test_oracle_sampling_of_generated_maps_matches_texels_at_centres():
    """at texel centres bilinear sampling returns the texel: ties the sampler's (row, column) convention to the maps"""
    n, ids = 128, [0, 1]
    g = H.oracle_generator(n, ids)
    g.update_all(UPDATE_DELTA)
    disp = np.stack([g.displacement(i) for i in range(2)])
    norm = np.stack([g.normal(i) for i in range(2)])
    tile = cascade_preset(0)["tile_length"][0]
    sc = np.array([[1 / tile, 1 / tile, 1.0, 1.0]], np.float32)
    cols, rows = np.array([0, 5, 127, 64]), np.array([0, 9, 127, 1])
    xz = np.stack([(cols + 0.5) / n * tile, (rows + 0.5) / n * tile], axis=1).astype(np.float32)
    o = O.sample_surface(disp[:1], norm[:1], sc, xz)
    want = disp[0].view(np.float16)[rows, cols, :3].astype(np.float32)
    assert np.abs(o["displacement"] - want).max() < 2e-3 * np.abs(want).max()


@pytest.mark.gpu
@pytest.mark.parametrize("n,ids", [(256, [0, 1, 2]), (128, [0, 1, 2, 3, 4, 5, 6, 7]), (1024, [2])])
def test_hip_sampling_is_bit_exact_against_the_oracle(n, ids):
    from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator
    gen = WaveGenerator()
    gen.map_size = n
    gen.init_gpu(len(ids))
    params = [WaveCascadeParameters(**cascade_preset(ci)) for ci in ids]
    gen.run(UPDATE_DELTA, params, 40)   # enough ticks for foam to build up, so that the spawn mask is not empty
    sc = np.array([(1 / p.tile_length[0], 1 / p.tile_length[1], p.displacement_scale, 0.5 + 0.1 * i) for i, p in enumerate(params)], np.float32)
    xz = query_points(20000)
    got = gen.sample_surface(xz, sc)
    maps = [gen.get_maps(i) for i in range(len(ids))]
    want = O.sample_surface(np.stack([m[0] for m in maps]), np.stack([m[1] for m in maps]), sc, xz)
    for f in O.SURFACE_SAMPLE.names:
        assert np.array_equal(got[f], want[f]), f
    assert 0 < want["foam"].max()
    # fewer cascades than the context holds: only the first layers are summed
    got1 = gen.sample_surface(xz[:100], sc[:1])
    want1 = O.sample_surface(np.stack([maps[0][0]]), np.stack([maps[0][1]]), sc[:1], xz[:100])
    assert np.array_equal(got1["displacement"], want1["displacement"])


@pytest.mark.gpu
def test_hip_sampling_argument_errors():
    from godotoceanwaves_amd import WaveGenerator
    gen = WaveGenerator()
    gen.map_size = 128
    gen.init_gpu(2)
    xz = np.zeros((4, 2), np.float32)
    with pytest.raises(_lib.OceanWavesError) as e:
        gen.sample_surface(xz, np.ones((3, 4), np.float32))      # more cascades than the context holds
    assert e.value.status == _lib.OW_ERR_INVALID
    assert len(gen.sample_surface(np.zeros((0, 2), np.float32), np.ones((1, 4), np.float32))) == 0
