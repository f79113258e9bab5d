"""ow_run on the compact family goes out in TICK PAIRS (k_tick_pair_c, at 2048^2 k_tick_pair_c_split: pass 2 of one batch of at most 4 Mi texels and
pass 1 of the next batch of the run in one launch, the compact family's own item bodies) -- held to one launch per pass BITWISE below, too.
ow_run on a small batch: from the second tick on, pass 2 of tick k and pass 1 of tick k + 1 go out in ONE launch (k_tick_group_c_lp;
the two are independent, the scratch intermediate is double-buffered by tick parity) -- against the same ticks as one pair of
launches each.  Same lane code in the same order per texel, so the comparison is BITWISE; the golden 1000-frame loop
(tests/test_golden.py, BASELINE config C2) runs through the tick groups as well and holds them to the oracle's trajectory."""
import numpy as np
import pytest

import helpers as H
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator
from godotoceanwaves_amd.presets import UPDATE_DELTA, cascade_preset

pytestmark = pytest.mark.gpu


def make(n, ids, tick_groups, debug=False, forms=(None, None)):
    gen = WaveGenerator()
    gen.map_size = n
    gen.tick_groups = tick_groups
    gen.debug_f32 = debug
    gen.group_forms = forms
    gen.init_gpu(max(2, len(ids)))
    return gen, [WaveCascadeParameters(**cascade_preset(ci)) for ci in ids]


def same_maps(a, b, count):
    for i in range(count):
        da, na = a.get_maps(i)
        db, nb = b.get_maps(i)
        assert np.array_equal(da.view(np.uint16), db.view(np.uint16)), i
        assert np.array_equal(na.view(np.uint16), nb.view(np.uint16)), i


@pytest.mark.parametrize("forms", [(None, None), ("lp", "plain"), ("compact", "plain"), ("lp", "pipe"), ("compact", "pipe")], ids=lambda f: f"p1_{f[0]}-p2_{f[1]}")
@pytest.mark.parametrize("n,ids", [(256, [0, 1, 2, 3]), (256, [0, 1, 2, 3, 4, 5, 6, 7]), (512, [2]), (512, [0, 1, 2, 3]), (1024, [1])])
@pytest.mark.parametrize("frames", [2, 3, 4, 17, 40])
def test_tick_groups_equal_one_launch_pair_per_tick(n, ids, frames, forms):
    """The runtime picks the form of the groups' work items by batch size -- pass 1: layer-parallel items or k_pass1c-shaped 8-row items;
    pass 2: plain blocks (a block walks through the ticks of its columns) or pipelined ones (the block's two halves on alternate ticks,
    foam handed over through LDS; blocks at most one per CU).  Every combination is held to the same bits at every size
    (OW_FLAG_GROUP_P1_* / OW_FLAG_GROUP_P2_* of ow_config; (None, None) = the runtime's own choice)."""
    a, pa = make(n, ids, True, forms=forms)
    b, pb = make(n, ids, False)
    a.run(UPDATE_DELTA, pa, frames)
    b.run(UPDATE_DELTA, pb, frames)
    a.sync(); b.sync()
    assert a.last_kernel_family() == ("tick_groups_compact" if frames >= 3 else "layer_parallel_compact")
    assert b.last_kernel_family() == "layer_parallel_compact"
    same_maps(a, b, len(ids))
    for x, y in zip(pa, pb):
        assert x.time == y.time and x.foam_grow_rate == y.foam_grow_rate and x.foam_decay_rate == y.foam_decay_rate
        assert not x.should_generate_spectrum
    # ... and the state carries on seamlessly on either path (foam plane, times, scratch halves)
    a.update_all(UPDATE_DELTA, pa); b.update_all(UPDATE_DELTA, pb)
    a.run(UPDATE_DELTA, pa, 6); b.run(UPDATE_DELTA, pb, 6)
    a.sync(); b.sync()
    same_maps(a, b, len(ids))


@pytest.mark.parametrize("n,ids", [(1024, [1, 2]), (1024, [0, 1, 2]), (1024, [0, 1, 2, 3]), (512, [0, 1, 2, 3, 4, 5, 6, 7]),
                                   (1024, [0, 1, 2, 3, 4]), (1024, [0, 1, 2, 3, 4, 5, 6]),  # two batches per tick: 3 + 2, 4 + 3
                                   (1024, [0, 1, 2, 3, 4, 5, 6, 7]),  # two batches of four, the stream cascade-major (each batch through up to 64 ticks before the other)
                                   # 2048^2 (k_tick_pair_c_split, round 4): one cascade per batch; pass 2's 16-wave blocks beside pass-1 blocks that
                                   # hold the two 4-row split-plan items of an 8-row unit, each with its own LDS arrival counter as its barrier
                                   (2048, [1]), (2048, [0, 2]), (2048, [0, 1, 2, 3])])
@pytest.mark.parametrize("frames", [2, 3, 4, 9])
def test_tick_pairs_equal_one_launch_per_pass(n, ids, frames):
    a, pa = make(n, ids, True)
    b, pb = make(n, ids, False)
    a.run(UPDATE_DELTA, pa, frames)
    b.run(UPDATE_DELTA, pb, frames)
    a.sync(); b.sync()
    assert a.last_kernel_family() == ("tick_pairs_compact" if frames >= 3 else "compact")
    assert b.last_kernel_family() == "compact"
    if frames >= 3:
        assert a.tick_group_depth() == 1
    same_maps(a, b, len(ids))
    for x, y in zip(pa, pb):
        assert x.time == y.time and x.foam_grow_rate == y.foam_grow_rate and x.foam_decay_rate == y.foam_decay_rate
    # the state carries on seamlessly on either path: the reference's schedule, a run on fewer cascades, another full run
    for gen, p in ((a, pa), (b, pb)):
        gen.update(UPDATE_DELTA, p)
        for _ in range(len(ids) - 1):
            gen._process(0.0)
        gen.run(UPDATE_DELTA, p[:2], 5)
        gen.run(UPDATE_DELTA, p, 6)
        gen.sync()
    same_maps(a, b, len(ids))
    assert [x.time for x in pa] == [y.time for y in pb]


@pytest.mark.parametrize("n,ids,frames", [(1024, [0, 1, 2, 3, 4], 150), (2048, [0, 1, 2], 70), (1024, [0, 1, 2, 3, 4, 5, 6, 7], 67)])
def test_cascade_major_stream_equals_one_launch_per_pass_across_block_boundaries(n, ids, frames):
    """A tick of several batches goes out cascade-major in blocks of 64 ticks (ow_runtime.hip run_tick_pairs): batch 0 through 64 ticks, then
    batch 1 through the same 64, ...  Runs longer than a block, with a partial last block, unequal batches (3 + 2) and the split-plan kernels,
    against the same ticks one launch per pass: same bits, same times."""
    a, pa = make(n, ids, True)
    b, pb = make(n, ids, False)
    a.run(UPDATE_DELTA, pa, frames)
    b.run(UPDATE_DELTA, pb, frames)
    a.sync(); b.sync()
    assert a.last_kernel_family() == "tick_pairs_compact" and b.last_kernel_family() == "compact"
    same_maps(a, b, len(ids))
    for x, y in zip(pa, pb):
        assert x.time == y.time and x.foam_grow_rate == y.foam_grow_rate and x.foam_decay_rate == y.foam_decay_rate


def test_tick_pairs_keep_the_debug_channels_and_match_the_oracle():
    n, ids, frames = 512, [0, 1, 2, 3, 4, 5, 6, 7], 4
    gen, params = make(n, ids, True, debug=True)
    og = H.oracle_generator(n, ids)
    gen.run(UPDATE_DELTA, params, frames)
    for _ in range(frames):
        og.update_all(UPDATE_DELTA)
    gen.sync()
    assert gen.last_kernel_family() == "tick_pairs_compact"
    for i in (0, 7):
        f32, ref = gen.get_maps_f32(i), og.f32(i)
        for c, name in enumerate(H.CHANNELS):
            if name == "foam":
                assert np.abs(f32[..., c] - ref[..., c]).max() <= H.TOL_FOAM_ABS, (i, name)
            else:
                assert H.relmax(f32[..., c], ref[..., c]) < H.TOL_F32, (i, name)
        disp, norm = gen.get_maps(i)
        assert H.quantisation_exact(f32, disp, norm)


def test_timing_as_launched_keeps_the_merged_launches():
    """timing mode 2: tick groups / tick pairs stay on and every such launch is timed; mode 1 puts ow_run back on one launch per pass"""
    for n, ids, fam, depth in ((1024, [0, 1], "tick_pairs_compact", 1), (256, [0, 1, 2, 3], "tick_groups_compact", 12)):
        gen, p = make(n, ids, True)
        gen.timing(2)
        gen.run(UPDATE_DELTA, p, 34)
        gen.sync()
        ms, launches = gen.timing_read_launches()
        assert gen.last_kernel_family() == fam and gen.tick_group_depth() == depth
        assert launches == -(-33 // depth) + 1 and 0.0 < ms < 5.0
        p1, p2, pairs = gen.timing_read()
        assert pairs == 1 and p1 > 0 and p2 > 0   # the run's first tick took the ordinary path, timed per pass
        gen.timing(True)
        gen.run(UPDATE_DELTA, p, 5)
        gen.sync()
        assert gen.last_kernel_family() in ("compact", "layer_parallel_compact")
        assert gen.timing_read()[2] == 5 and gen.timing_read_launches()[1] == 0
        gen.timing(False)


def test_tick_groups_match_the_oracle_and_keep_the_debug_channels():
    n, ids, frames = 256, [0, 1, 2, 3], 6
    gen, params = make(n, ids, True, debug=True)
    og = H.oracle_generator(n, ids)
    gen.run(UPDATE_DELTA, params, frames)
    for _ in range(frames):
        og.update_all(UPDATE_DELTA)
    gen.sync()
    assert gen.last_kernel_family() == "tick_groups_compact"
    for i in range(len(ids)):
        assert params[i].time == og.params[i].time
        f32, ref = gen.get_maps_f32(i), og.f32(i)
        for c, name in enumerate(H.CHANNELS):
            if name == "foam":
                assert np.abs(f32[..., c] - ref[..., c]).max() <= H.TOL_FOAM_ABS, (i, name)
            else:
                assert H.relmax(f32[..., c], ref[..., c]) < H.TOL_F32, (i, name)
        disp, norm = gen.get_maps(i)
        assert H.quantisation_exact(f32, disp, norm)


def test_a_dirty_record_or_a_large_batch_stays_off_the_tick_groups():
    """live parameter edits regenerate the spectrum through the ordinary first tick; batches outside the small-batch family never enter"""
    n, ids = 256, [0, 1, 2, 3]
    a, pa = make(n, ids, True)
    b, pb = make(n, ids, False)
    a.run(UPDATE_DELTA, pa, 4); b.run(UPDATE_DELTA, pb, 4)
    pa[2].wind_speed = 7.0; pb[2].wind_speed = 7.0
    a.run(UPDATE_DELTA, pa, 4); b.run(UPDATE_DELTA, pb, 4)
    a.sync(); b.sync()
    assert a.last_kernel_family() == "tick_groups_compact"
    same_maps(a, b, len(ids))
    for n_big, ids_big in ((2048, [0]), (2048, [0, 1, 2])):  # 2048^2: tick pairs of one cascade per batch (k_tick_pair_c_split), never tick groups
        big, pbig = make(n_big, ids_big, True)
        big.run(UPDATE_DELTA, pbig, 4)
        big.sync()
        assert big.last_kernel_family() == "tick_pairs_compact" and big.last_batch_cascades() == 1 and big.tick_group_depth() == 1
        big.free()
    # 1024^2 x 8: two full-size batches of four, the stream in cascade-major order (round 3's half-size batches are gone: a batch now shares
    # the Infinity Cache with itself one tick later, not with the other batch)
    big, pbig = make(1024, list(range(8)), True)
    big.run(UPDATE_DELTA, pbig, 4)
    big.sync()
    assert big.last_kernel_family() == "tick_pairs_compact" and big.last_batch_cascades() == 4
    big.free()


def test_tick_groups_interleaved_with_the_reference_schedule_and_changing_counts():
    """run() on all four cascades, then the reference's own schedule (update + one cascade per frame, one left for the flush), then run()
    on the first two cascades only, then a zero-frame run: the same calls with tick groups off give the same bits everywhere"""
    n, ids = 256, [0, 1, 2, 3]

    def drive(tick_groups):
        gen, p = make(n, ids, tick_groups)
        gen.run(UPDATE_DELTA, p, 7)
        gen.update(UPDATE_DELTA, p)
        for _ in range(3):
            gen._process(0.0)                      # cascade 0 stays armed: flushed by the first tick of the next run
        gen.run(UPDATE_DELTA, p[:2], 6)            # fewer cascades than the context holds
        gen.run(UPDATE_DELTA, p, 0)                # nothing
        gen.run(UPDATE_DELTA, p, 9)
        gen.sync()
        return gen, p

    a, pa = drive(True)
    b, pb = drive(False)
    assert a.last_kernel_family() == "tick_groups_compact" and b.last_kernel_family() == "layer_parallel_compact"
    same_maps(a, b, len(ids))
    assert [x.time for x in pa] == [y.time for y in pb]


def test_group_depth_follows_the_run_and_the_scratch_grows_on_first_use():
    """the depth of a tick group follows the cascade count of THAT run (512^2: sixteen ticks per launch for one cascade,
    eight for two, four beyond), not the largest count the context could ever see; and the deeper scratch intermediate the merged launches need is
    allocated by the first ow_run that uses them, mid-stream, without disturbing the simulation (same bits as a context that never
    merges)."""
    n, ids = 512, list(range(6))
    a, pa = make(n, ids, True)
    b, pb = make(n, ids, False)
    for gen, p in ((a, pa), (b, pb)):
        gen.update_all(UPDATE_DELTA, p)              # ordinary ticks first: one batch of scratch
        gen.update(UPDATE_DELTA, p)
        gen._process(0.0)
    a.run(UPDATE_DELTA, pa[:2], 20)                  # (flushes the leftovers, then the groups: the scratch grows here)
    assert a.last_kernel_family() == "tick_groups_compact" and a.tick_group_depth() == 8
    a.run(UPDATE_DELTA, pa, 20)
    assert a.last_kernel_family() == "tick_groups_compact" and a.tick_group_depth() == 4
    a.run(UPDATE_DELTA, pa[:1], 11)
    assert a.tick_group_depth() == 16
    b.run(UPDATE_DELTA, pb[:2], 20)
    b.run(UPDATE_DELTA, pb, 20)
    b.run(UPDATE_DELTA, pb[:1], 11)
    a.sync(); b.sync()
    same_maps(a, b, len(ids))
    assert [x.time for x in pa] == [y.time for y in pb]
