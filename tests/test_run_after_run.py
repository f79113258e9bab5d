"""ow_run after ow_run (include/ocean_waves.h, ow_runtime.hip run_impl): the last launch of a run that followed a run with the same delta and cascade
count also carries pass 1 of what the next such run starts with -- the first tick group, or the next tick of the batch the cascade-major pair stream
ended on -- and the next run resumes in the middle of the stream: no ordinary first tick, no half-filled launches at the ends.  Whatever happens in
between, the maps are BITWISE those of a context that never merges anything (OW_FLAG_NO_TICK_GROUPS)."""
import numpy as np
import pytest

from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator
from godotoceanwaves_amd.presets import UPDATE_DELTA, cascade_preset

pytestmark = pytest.mark.gpu

# tick groups (the layer-parallel compact family) and multi-batch tick pairs (cascade-major stream): the shapes round 5 left with a per-call cost
SHAPES = [(256, 4, "tick_groups_compact"), (256, 1, "tick_groups_compact"), (512, 2, "tick_groups_compact"), (1024, 1, "tick_groups_compact"),
          (1024, 8, "tick_pairs_compact"), (1024, 5, "tick_pairs_compact"), (2048, 2, "tick_pairs_compact")]


def make(n, count, merge=True):
    gen = WaveGenerator()
    gen.map_size, gen.tick_groups = n, merge
    gen.init_gpu(max(2, count))
    return gen, [WaveCascadeParameters(**cascade_preset(i)) for i in range(count)]


def same(a, b, count):
    a.sync(); b.sync()
    for i in range(count):
        da, na = a.get_maps(i)
        db, nb = b.get_maps(i)
        assert np.array_equal(da.view(np.uint16), db.view(np.uint16)), i
        assert np.array_equal(na.view(np.uint16), nb.view(np.uint16)), i


@pytest.mark.parametrize("n,count,family", SHAPES, ids=[f"{n}x{c}" for n, c, _ in SHAPES])
def test_back_to_back_runs_resume_in_the_middle_of_the_stream(n, count, family):
    a, pa = make(n, count)
    b, pb = make(n, count, merge=False)
    a.run(UPDATE_DELTA, pa, 20); b.run(UPDATE_DELTA, pb, 20)
    assert a.lookahead_stats() == (0, 0)           # a one-shot run leaves nothing in the queue
    same(a, b, count)
    a.run(UPDATE_DELTA, pa, 20); b.run(UPDATE_DELTA, pb, 20)
    hits, spec = a.lookahead_stats()
    assert hits == 0 and spec == 1                 # the second run knows it follows one: its last launch works ahead
    same(a, b, count)                              # (reading the maps in between disturbs nothing)
    for frames in (20, 20, 7, 1, 33, 70, 2, 20):   # other lengths resume as well (a shorter run uses a part of the group computed ahead)
        a.run(UPDATE_DELTA, pa, frames); b.run(UPDATE_DELTA, pb, frames)
        assert a.last_kernel_family() == family
    hits2, spec2 = a.lookahead_stats()
    assert hits2 >= 8 and spec2 == 9               # every one of the eight resumed, every one worked ahead again
    same(a, b, count)
    assert [p.time for p in pa] == [p.time for p in pb]
    assert [p.foam_grow_rate for p in pa] == [p.foam_grow_rate for p in pb]
    a.free(); b.free()


@pytest.mark.parametrize("n,count,family", [SHAPES[0], SHAPES[4], SHAPES[6]], ids=["256x4", "1024x8", "2048x2"])
def test_everything_that_may_come_between_two_runs(n, count, family):
    a, pa = make(n, count)
    b, pb = make(n, count, merge=False)

    def both(f):
        f(a, pa); f(b, pb)

    run = lambda k, d=UPDATE_DELTA: both(lambda g, p: g.run(d, p, k))
    run(12); run(12); run(12)
    h0 = a.lookahead_stats()[0]
    assert h0 >= 1
    # a parameter pass 1 does not depend on, edited the reference's way (dirty flag: wave_cascade_parameters.gd:32-33): the spectrum is resident, the run resumes
    def whitecap(g, p):
        p[0].whitecap = 0.77
    both(whitecap); run(12)
    assert a.lookahead_stats()[0] > h0 and a.spectrum_stats()[1] >= 1
    same(a, b, count)
    # a real edit: the run starts the ordinary way (spectrum regenerated), the one after it resumes again
    def wind(g, p):
        p[count - 1].wind_speed = 11.0
    h1 = a.lookahead_stats()[0]
    both(wind); run(12)
    assert a.lookahead_stats()[0] == h1
    run(12); run(12)
    assert a.lookahead_stats()[0] > h1
    same(a, b, count)
    # another delta, and back; a tick issued by the caller in between; the reference's schedule with a leftover; fewer cascades; a restored foam plane
    run(9, 0.03); run(9, 0.03); run(9, 0.03); run(9)
    both(lambda g, p: g.update_all(UPDATE_DELTA, p)); run(10); run(10)
    def reference(g, p):
        g.update(UPDATE_DELTA, p)
        for _ in range(max(0, count - 1)):
            g._process(0.0)
    both(reference); run(10); run(10); run(10)
    if count > 1:
        both(lambda g, p: g.run(UPDATE_DELTA, p[:count - 1], 8)); both(lambda g, p: g.run(UPDATE_DELTA, p[:count - 1], 8)); both(lambda g, p: g.run(UPDATE_DELTA, p[:count - 1], 8))
    run(10); run(10)
    a.sync()
    saved = a.get_maps(0)[1].copy()
    both(lambda g, p: g.set_normal_map(0, saved)); run(10); run(10)
    # a zero-frame run and a readback between runs
    run(0); run(10)
    a.readback_begin([0]); a.readback_wait(0); run(10); run(10)
    same(a, b, count)
    assert a.last_kernel_family() == family
    assert [p.time for p in pa] == [p.time for p in pb]
    a.free(); b.free()


def test_a_fault_between_two_runs_drops_what_was_computed_ahead():
    from godotoceanwaves_amd._lib import OceanWavesError, OW_ERR_HIP
    n, count = 256, 4
    a, pa = make(n, count)
    b, pb = make(n, count, merge=False)
    for _ in range(3):
        a.run(UPDATE_DELTA, pa, 12); b.run(UPDATE_DELTA, pb, 12)
    hits = a.lookahead_stats()[0]
    a.debug_inject_fault(2)                        # the status word as a faulting launch IN FLIGHT would leave it
    with pytest.raises(OceanWavesError) as e:
        a.sync()
    assert e.value.status == OW_ERR_HIP
    a.run(UPDATE_DELTA, pa, 12); b.run(UPDATE_DELTA, pb, 12)   # starts the ordinary way: nothing poisoned is resumed from
    assert a.lookahead_stats()[0] == hits
    same(a, b, count)
    a.free(); b.free()
