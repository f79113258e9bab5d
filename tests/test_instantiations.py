"""The FP32-channel evidence (<= 1e-4 against the oracle, tests/test_gpu_parity.py) is taken through OW_FLAG_DEBUG_F32, i.e. on the `F32 = true`
instantiations of the pass-2 kernels, while what a caller runs -- and what bench.py times -- are the `F32 = false` ones.
The two differ by the debug stores only; this file holds them to the same BITS on every BASELINE configuration: a debug context and a
production context side by side through the same calls, every RGBA16F texel of both maps (the foam state is normal.a) bit-identical -- through
ow_run's merged launches (tick groups / tick pairs), through one launch per pass, and through the reference's call-by-call schedule."""
import numpy as np
import pytest

import helpers as H
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator
from godotoceanwaves_amd.presets import UPDATE_DELTA, cascade_preset

pytestmark = pytest.mark.gpu

BASELINE_CONFIGS = [(256, 4), (1024, 4), (1024, 8), (2048, 4)]   # C2, C3 (headline), C4 (per node), C5


def make(n, count, debug, **kw):
    gen = WaveGenerator()
    gen.map_size, gen.debug_f32 = n, debug
    for k, v in kw.items():
        setattr(gen, k, v)
    gen.init_gpu(max(2, count))
    return gen, [WaveCascadeParameters(**cascade_preset(i)) for i in range(count)]


def same_bits(a, b, count, where):
    a.sync(); b.sync()
    for i in range(count):
        da, na = a.get_maps(i)
        db, nb = b.get_maps(i)
        assert np.array_equal(da.view(np.uint16), db.view(np.uint16)), (where, i, "displacement")
        assert np.array_equal(na.view(np.uint16), nb.view(np.uint16)), (where, i, "normal / foam")
        assert float(np.abs(da[..., :3].astype(np.float32)).max()) > 0.0   # real maps, not zeros


@pytest.mark.parametrize("n,count", BASELINE_CONFIGS, ids=lambda v: str(v))
def test_debug_and_production_instantiations_leave_the_same_bits(n, count):
    dbg, pd = make(n, count, True)
    prod, pp = make(n, count, False)
    dbg.run(UPDATE_DELTA, pd, 3); prod.run(UPDATE_DELTA, pp, 3)          # spectrum tick + the first merged launches
    assert dbg.last_kernel_family() == prod.last_kernel_family()
    assert prod.last_kernel_family() in ("tick_groups_compact", "tick_pairs_compact")   # the launches bench.py times
    same_bits(dbg, prod, count, "ow_run(3)")
    frames = 5 if n >= 2048 else 12
    dbg.run(UPDATE_DELTA, pd, frames); prod.run(UPDATE_DELTA, pp, frames)
    same_bits(dbg, prod, count, f"ow_run({frames}) more")
    for _ in range(3):                                                      # tick by tick: the look-ahead's launches
        dbg.update_all(UPDATE_DELTA, pd); prod.update_all(UPDATE_DELTA, pp)
    same_bits(dbg, prod, count, "update_all x 3")
    dbg.update(UPDATE_DELTA, pd); prod.update(UPDATE_DELTA, pp)            # the reference's schedule
    for _ in range(count):
        dbg._process(0.0); prod._process(0.0)
    same_bits(dbg, prod, count, "update + process")
    # ... and the debug image IS what the maps quantise: RTE of its channels, bit for bit (ties tests/test_gpu_parity.py's FP32 evidence to these maps)
    for i in (0, count - 1):
        assert H.quantisation_exact(dbg.get_maps_f32(i), *prod.get_maps(i)), i


@pytest.mark.parametrize("n,count", [(1024, 4), (2048, 4), (256, 4)], ids=lambda v: str(v))
def test_one_launch_per_pass_instantiations_too(n, count):
    dbg, pd = make(n, count, True, tick_groups=False)
    prod, pp = make(n, count, False, tick_groups=False)
    dbg.run(UPDATE_DELTA, pd, 4); prod.run(UPDATE_DELTA, pp, 4)
    assert prod.last_kernel_family() in ("compact", "layer_parallel_compact")
    same_bits(dbg, prod, count, "one launch per pass")
