"""ow_update_all's adaptive look-ahead (include/ocean_waves.h): once two consecutive calls have come with the same delta, a call launches a
SPECULATED pass 1 of the next tick together with its own pass 2, and the next call -- if what it is given matches the speculation bit for
bit -- costs one merged launch instead of two.  Whatever happens (hits, misses, edits, other calls in between), the maps are BITWISE those
of a context that never merges anything (OW_FLAG_NO_TICK_GROUPS), tick by tick."""
import numpy as np
import pytest

import helpers as H
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator
from godotoceanwaves_amd.presets import UPDATE_DELTA, cascade_preset

pytestmark = pytest.mark.gpu


def make(n, ids, merge=True, debug=False, run_as_calls=False):
    gen = WaveGenerator()
    gen.map_size, gen.tick_groups, gen.debug_f32, gen.run_as_calls = n, merge, debug, run_as_calls
    gen.init_gpu(max(2, len(ids)))
    return gen, [WaveCascadeParameters(**cascade_preset(ci)) for ci in ids]


def same(a, b, count):
    a.sync(); b.sync()
    for i in range(count):
        da, na = a.get_maps(i)
        db, nb = b.get_maps(i)
        assert np.array_equal(da.view(np.uint16), db.view(np.uint16)), i
        assert np.array_equal(na.view(np.uint16), nb.view(np.uint16)), i


@pytest.mark.parametrize("n,ids", [(1024, [0, 1, 2, 3]), (256, [0, 1, 2, 3]), (2048, [1]), (512, [0, 1, 2, 3, 4, 5, 6, 7]), (1024, [2]), (1024, [0, 2])])
def test_regular_cadence_hits_and_equals_one_launch_per_pass(n, ids):
    a, pa = make(n, ids)
    b, pb = make(n, ids, merge=False)
    for k in range(12):
        a.update_all(UPDATE_DELTA, pa)
        b.update_all(UPDATE_DELTA, pb)
        if k in (2, 7):
            same(a, b, len(ids))   # reading the maps in between does not disturb the speculation
    same(a, b, len(ids))
    hits, spec = a.lookahead_stats()
    assert hits == 10   # call 1: nothing; call 2: computes ahead (its own pass 1 was not); calls 3 .. 12: hits
    # launches that carried work for later ticks -- the pair kernel one tick per launch: calls 2 .. 12; the group kernel (layer-parallel
    # family) as many ticks as the delta has repeated, up to four: call 2 (one tick), call 3 (two), call 5 (four), call 9
    assert spec == (4 if a.last_kernel_family() == "layer_parallel_compact" else 11)
    assert b.lookahead_stats() == (0, 0)
    assert [p.time for p in pa] == [p.time for p in pb]
    assert a.last_kernel_family() == b.last_kernel_family()


@pytest.mark.parametrize("n", [1024, 256])   # (the pair kernel: one tick ahead; the group kernel: up to four)
def test_jittering_deltas_never_arm_it_and_a_changed_delta_is_a_miss(n):
    ids = [0, 1, 2]
    a, pa = make(n, ids)
    b, pb = make(n, ids, merge=False)
    for k in range(8):   # water.gd's rate limiter passes the elapsed time: no two deltas alike
        d = UPDATE_DELTA * (1.0 + 0.01 * k)
        a.update_all(d, pa); b.update_all(d, pb)
    same(a, b, 3)
    assert a.lookahead_stats() == (0, 0)
    # 0.02 x 3: -, speculate, hit + speculate; 0.05 x 3: miss (the speculated time is wrong), speculate, hit -- and NO speculation: the previous
    # run of equal deltas was three updates long, so a fourth 0.05 is not assumed; 0.02: nothing outstanding, nothing thrown away
    for d in (0.02, 0.02, 0.02, 0.05, 0.05, 0.05, 0.02):
        a.update_all(d, pa); b.update_all(d, pb)
    same(a, b, 3)
    hits, spec = a.lookahead_stats()
    assert hits == 2 and spec == 3
    # a caller whose delta changes every second update is left alone altogether after its first change
    for k in range(12):
        d = 0.05 if (k // 2) % 2 == 0 else 0.02   # (the call before was a 0.02: the pattern starts with a change)
        a.update_all(d, pa); b.update_all(d, pb)
    same(a, b, 3)
    hits2, spec2 = a.lookahead_stats()
    assert hits2 == hits and spec2 <= spec + 1


@pytest.mark.parametrize("n,ids", [(512, [0, 1, 2, 3, 4, 5, 6, 7]), (256, [0, 1, 2, 3, 4]), (512, [4, 5])])
def test_everything_that_invalidates_a_speculation(n, ids):
    """(512^2 x 8: the pair kernel, one tick ahead; the other two: the group kernel, whose queue of up to four ticks is cut short by each event)"""
    a, pa = make(n, ids)
    b, pb = make(n, ids, merge=False)

    def both(f):
        f(a, pa); f(b, pb)

    tick = lambda g, p: g.update_all(UPDATE_DELTA, p)
    for _ in range(4):
        both(tick)
    h0 = a.lookahead_stats()[0]
    assert h0 == 2
    # a tile length edited between two calls (pass 1 depends on it): the dirty flag sends the tick down the ordinary path
    def edit_tile(g, p):
        p[-1].tile_length = (41.0, 43.0)
    both(edit_tile); both(tick); both(tick); both(tick)
    # a parameter pass 1 does NOT depend on, edited without the dirty flag (the C caller's way): still a hit, pass 2 sees the new value
    def edit_whitecap(g, p):
        p[1]._whitecap = 0.9
    both(edit_whitecap); both(tick)
    # fewer cascades, the reference's schedule in between, a run, a restored foam state
    fewer = max(1, len(ids) - 3)
    both(lambda g, p: g.update_all(UPDATE_DELTA, p[:fewer]))
    both(lambda g, p: g.update_all(UPDATE_DELTA, p[:fewer]))
    both(lambda g, p: g.update_all(UPDATE_DELTA, p[:fewer]))
    def reference_schedule(g, p):
        g.update(UPDATE_DELTA, p)
        for _ in range(min(3, len(ids))):
            g._process(0.0)
    both(reference_schedule); both(tick); both(tick); both(tick)
    both(lambda g, p: g.run(UPDATE_DELTA, p, 7)); both(tick); both(tick)
    saved = a.get_maps(1)[1].copy()
    both(lambda g, p: g.set_normal_map(1, saved)); both(tick); both(tick)
    # a changed delta in the middle of a queue of ticks computed ahead, and back
    for d in (UPDATE_DELTA,) * 4 + (0.03,) * 5 + (UPDATE_DELTA,) * 6:
        both(lambda g, p: g.update_all(d, p))
    same(a, b, len(ids))
    assert [p.time for p in pa] == [p.time for p in pb]
    assert a.lookahead_stats()[0] > h0 + 4


def test_two_batch_ticks_and_pinned_families_stay_on_the_ordinary_path():
    a, pa = make(1024, [0, 1, 2, 3, 4])          # 3 + 2: a second batch would need its own two intermediates
    for _ in range(5):
        a.update_all(UPDATE_DELTA, pa)
    a.sync()
    assert a.lookahead_stats() == (0, 0)
    g = WaveGenerator()
    g.map_size, g.kernels = 512, "standard"        # the four-layer kernels have no merged form
    g.init_gpu(2)
    p = [WaveCascadeParameters(**cascade_preset(i)) for i in range(2)]
    for _ in range(5):
        g.update_all(UPDATE_DELTA, p)
    g.sync()
    assert g.lookahead_stats() == (0, 0) and g.last_kernel_family() == "standard"


def test_run_as_calls_is_the_tick_by_tick_caller_without_the_host_round_trips():
    n, ids = 1024, [0, 1, 2, 3]
    a, pa = make(n, ids, run_as_calls=True)
    b, pb = make(n, ids, merge=False)
    a.run(UPDATE_DELTA, pa, 40); b.run(UPDATE_DELTA, pb, 40)
    same(a, b, 4)
    hits, spec = a.lookahead_stats()
    assert hits == 38 and spec == 39 and a.last_kernel_family() == "compact"


def test_ticks_through_the_look_ahead_match_the_oracle():
    n, ids = 1024, [0, 2]
    gen, params = make(n, ids, debug=True)
    og = H.oracle_generator(n, ids)
    for _ in range(5):
        gen.update_all(UPDATE_DELTA, params)
        og.update_all(UPDATE_DELTA)
    gen.sync()
    assert gen.lookahead_stats()[0] == 3
    for i in range(2):
        assert params[i].time == og.params[i].time
        f32, ref = gen.get_maps_f32(i), og.f32(i)
        for c, name in enumerate(H.CHANNELS):
            if name == "foam":
                assert np.abs(f32[..., c] - ref[..., c]).max() <= H.TOL_FOAM_ABS, (i, name)
            else:
                assert H.relmax(f32[..., c], ref[..., c]) < H.TOL_F32, (i, name)
        disp, norm = gen.get_maps(i)
        assert H.quantisation_exact(f32, disp, norm)


def test_at_2048_only_the_same_cascade_is_computed_ahead():
    """pass 1 of ANOTHER cascade beside pass 2 loses at 2048^2 (the tick-major pairing): the reference schedule prefetches nothing there, except
    across updates where the next cascade is the same one (a context of one cascade)"""
    a, pa = make(2048, [0, 1])
    b, pb = make(2048, [0, 1], merge=False)
    one, p1 = make(2048, [2])
    for _ in range(4):
        for g, p in ((a, pa), (b, pb), (one, p1)):
            g.update(UPDATE_DELTA, p)
            while g.pass_num_cascades_remaining:
                g._process(0.0)
    same(a, b, 2)
    assert a.lookahead_stats() == (0, 0)
    assert one.lookahead_stats() == (2, 3)   # update 2 (its delta repeats update 1's) computes ahead for update 3; updates 3 and 4 hit


@pytest.mark.parametrize("n,count", [(1024, 4), (256, 4), (512, 3)])
def test_the_reference_schedule_prefetches_the_next_armed_cascade(n, count):
    """ow_update + one ow_process per frame (wave_generator.gd:56-63,90-109): the launch of cascade i carries pass 1 of cascade i - 1, whose armed
    record is KNOWN; only the step to the next update's first cascade is a guess (time + delta, once the deltas repeat).  Bitwise the maps of a
    context that never merges, and every ow_process after the first of the second update is a hit."""
    ids = list(range(count))
    a, pa = make(n, ids)
    b, pb = make(n, ids, merge=False)
    updates = 6
    for _ in range(updates):
        for g, p in ((a, pa), (b, pb)):
            g.update(UPDATE_DELTA, p)
            while g.pass_num_cascades_remaining:
                g._process(0.0)
    same(a, b, count)
    hits, spec = a.lookahead_stats()
    # update 1: every spectrum is generated (ordinary path); from update 2 on every ow_process is a hit: ow_update itself launches pass 1 of the
    # cascades the ow_process calls will take (round 5), or the previous update's last ow_process has guessed them
    assert hits == (updates - 1) * count
    # launches that carried pass 1 for later ones: one in four where a single launch takes the next four of the caller's launches
    assert hits // 4 <= spec <= hits + 1
    assert [p.time for p in pa] == [p.time for p in pb]


def test_reference_schedule_with_jitter_live_edits_and_leftovers():
    n, ids = 512, [0, 1, 2, 3]
    a, pa = make(n, ids)
    b, pb = make(n, ids, merge=False)

    def drive(g, p):
        for k in range(9):
            g.update(UPDATE_DELTA * (1.0 + 0.003 * (k % 3)), p)          # no two updates in a row alike: no guess across updates
            drain = 4 if k % 4 != 2 else 2                               # every fourth update leaves two cascades for the next update's flush
            for j in range(drain):
                if k == 5 and j == 1:
                    p[2].tile_length = (33.0, 35.0)                      # a live edit of the cascade that has just been prefetched: dirty -> ordinary path
                if k == 6 and j == 2:
                    p[1]._whitecap = 0.7                                  # pass 2 only: the prefetched pass 1 stays valid
                g._process(0.0)
        g.update(UPDATE_DELTA, p)
        while g.pass_num_cascades_remaining:
            g._process(0.0)

    drive(a, pa); drive(b, pb)
    same(a, b, 4)
    assert [p.time for p in pa] == [p.time for p in pb]
    assert a.lookahead_stats()[0] >= 12


def test_run_as_reference_schedule_is_update_plus_one_process_per_cascade():
    n, ids = 256, [0, 1, 2, 3]
    a, pa = make(n, ids)
    a.free()
    a = WaveGenerator()
    a.map_size, a.run_as_reference = n, True
    a.init_gpu(4)
    b, pb = make(n, ids, merge=False)
    a.run(UPDATE_DELTA, pa, 30); b.run(UPDATE_DELTA, pb, 30)
    same(a, b, 4)
    hits, spec = a.lookahead_stats()
    assert hits == 29 * 4 and a.pass_num_cascades_remaining == 0   # (all but the first update's, which generates the spectra)


@pytest.mark.parametrize("n,count", [(256, 4), (256, 1), (512, 2), (512, 8), (1024, 1), (1024, 3), (256, 8), (2048, 1), (1024, 4), (1024, 8), (2048, 2)])  # = fuzz_schedule.CONFIGS
def test_random_schedules_hold_the_bits_of_a_context_that_never_merges(n, count):
    """scripts/fuzz_schedule.py: random sequences of update_all (repeating and changing deltas), update + some or all of its process calls, short
    runs, live edits, fewer cascades, restored foam -- every merged launch shape against OW_FLAG_NO_TICK_GROUPS, bit for bit"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_schedule", os.path.join(os.path.dirname(__file__), "..", "scripts", "fuzz_schedule.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    assert (n, count) in fz.CONFIGS and len(fz.CONFIGS) == 11
    served = 0
    for seed in (11, 12, 13, 14, 15):   # 11 configurations x 5 fixed seeds = 55 schedules in the suite; the long run stays a script
        calls, hits = fz.schedule(n, count, seed, ops=30 if n >= 2048 else 40)
        served += hits
    assert served > 0   # (the schedules do reach the look-ahead)


@pytest.mark.parametrize("n,ids", [(1024, [0, 1, 2, 3]), (256, [0, 1, 2, 3]), (2048, [1])])
def test_a_fault_reported_while_work_is_queued_ahead_drops_the_queue(n, ids):
    """ADVICE r4 (ow_runtime.hip consume_status): a launch that faults may have written speculated pass-1 entries, too -- of cascades not yet
    enqueued, or of the same cascade one tick later.  Once the status word has been consumed nothing would refuse them any more, so the
    synchronising call that reports the fault also drops the queue: the ticks after it recompute their pass 1 (no hit on poisoned work) and
    land on the bits of a context that never speculates.  The fault is injected as the status word a faulting launch IN FLIGHT would leave
    (ow_debug_inject_fault bit 1); bit 0, a fault of the next batch, keeps the look-ahead off and cannot reach a speculating launch."""
    from godotoceanwaves_amd._lib import OceanWavesError, OW_ERR_HIP
    a, pa = make(n, ids)
    b, pb = make(n, ids, merge=False)
    for _ in range(4):   # armed: every call from the third on hits and computes ahead again
        a.update_all(UPDATE_DELTA, pa); b.update_all(UPDATE_DELTA, pb)
    hits0, spec0 = a.lookahead_stats()
    assert hits0 == 2 and spec0 >= 1
    a.debug_inject_fault(2)
    with pytest.raises(OceanWavesError) as e:
        a.sync()
    assert e.value.status == OW_ERR_HIP
    for i in range(len(ids)):   # the layers enqueued since the last synchronisation stay refused until recomputed
        with pytest.raises(OceanWavesError):
            a.get_maps(i)
    a.update_all(UPDATE_DELTA, pa); b.update_all(UPDATE_DELTA, pb)
    hits1, _ = a.lookahead_stats()
    assert hits1 == hits0, "the tick after a reported fault must not consume work computed ahead by the faulted launches"
    same(a, b, len(ids))   # (the injected word corrupted nothing: the recomputed tick is the never-merging context's, bit for bit)
    for _ in range(3):     # ... and the look-ahead arms again
        a.update_all(UPDATE_DELTA, pa); b.update_all(UPDATE_DELTA, pb)
    same(a, b, len(ids))
    assert a.lookahead_stats()[0] > hits1


@pytest.mark.parametrize("n,count", [(1024, 4), (256, 4), (512, 8)])
def test_the_scenes_cadence_update_launches_pass_1_and_the_flush_consumes_the_queue(n, count):
    """water.gd's rate limiter never issues the same delta twice, so nothing can be guessed across updates -- but nothing has to be: ow_update itself
    launches pass 1 of the cascades its ow_process calls will take (up to four, one launch), every ow_process is then a hit, and what an update
    leaves unprocessed (a frame rate below cascades x update rate) is flushed by the next update from the same queue, as ONE pass-2 launch in
    the kernel family a batch of that size takes anyway.  Bitwise the maps of a context that never merges (1024^2: a single cascade and a batch
    of two or three fall into different families -- the flush must not change the batch's), and every cascade of every update but the first
    (which generates the spectra) is served from work computed ahead."""
    ids = list(range(count))
    a, pa = make(n, ids)
    b, pb = make(n, ids, merge=False)
    drains = [count, count - 1, 1, 2, 0, count, 2, count - 2]
    for k, drain in enumerate(drains):
        d = UPDATE_DELTA * (1.0 + 0.013 * ((k * 7) % 5))   # no two updates in a row alike
        for g, p in ((a, pa), (b, pb)):
            g.update(d, p)
            for _ in range(drain):
                g._process(0.0)
        if k in (2, 5):
            same(a, b, count)   # reading the maps in between disturbs nothing
    for g, p in ((a, pa), (b, pb)):   # whatever the last update left is flushed by one more, which is then drained
        g.update(UPDATE_DELTA, p)
        while g.pass_num_cascades_remaining:
            g._process(0.0)
    same(a, b, count)
    assert [p.time for p in pa] == [p.time for p in pb]
    hits, spec = a.lookahead_stats()
    if count <= 4:
        assert hits == count * len(drains)        # updates 2 .. 9: every cascade, processed or flushed; update 1 generated the spectra
        assert spec == len(drains)                # one launch of pass 1 per update, nothing else carried work for later
    else:
        assert hits >= 4 * len(drains)            # (eight cascades: the queue holds four at a time; flushes of more than four take the ordinary path)
    assert b.lookahead_stats() == (0, 0)
