"""The C-ABI's contract around the hot path (include/ocean_waves.h, ABI version 2, kept in 3): the parameter records are COPIED by
ow_update (wave_generator.gd:108 keeps an Array reference; a C / C# caller's memory is only borrowed during the call), bad
records are refused on the way in (before anything of the caller's is changed, armed or launched), the exported setters'
clamps (wave_cascade_parameters.gd:15,20) hold for a caller that has no setters, and a device-side wait that gives up is
REPORTED through the status word instead of producing maps."""
import ctypes as C

import numpy as np
import pytest

import helpers as H
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, _lib
from godotoceanwaves_amd._lib import ow_cascade_params
from godotoceanwaves_amd.presets import UPDATE_DELTA, cascade_preset

pytestmark = pytest.mark.gpu


def raw_gen(n, cascades, **kw):
    gen = WaveGenerator()
    gen.map_size = n
    for k, v in kw.items():
        setattr(gen, k, v)
    gen.init_gpu(cascades)
    return gen, _lib.load()


def packed(ids):
    arr = (ow_cascade_params * len(ids))()
    for c, ci in zip(arr, ids):
        WaveCascadeParameters(**cascade_preset(ci))._pack(c)
    return arr


def test_update_copies_the_records_and_consumes_the_dirty_flag():
    """the caller's array is scribbled over right after ow_update: the one-cascade-per-frame drain still uses what was armed"""
    n, ids = 256, [0, 1, 2]
    ref, params = WaveGenerator(), [WaveCascadeParameters(**cascade_preset(ci)) for ci in ids]
    ref.map_size = n
    ref.init_gpu(len(ids))
    for _ in range(2):
        ref.update_all(UPDATE_DELTA, params)
    ref.sync()

    gen, L = raw_gen(n, len(ids))
    arr = packed(ids)
    for tick in range(2):
        _lib.check(L.ow_update(gen.context, UPDATE_DELTA, arr, len(ids)))
        assert [a.should_generate_spectrum for a in arr] == [0, 0, 0]          # consumed at arm time
        assert arr[1].time == params[1].time - (1 - tick) * UPDATE_DELTA        # advanced inside the caller's struct
        keep = bytes(arr)
        C.memset(arr, 0xFF, C.sizeof(arr))                                      # the caller's memory goes away
        got = ow_cascade_params()
        _lib.check(L.ow_get_cascade_params(gen.context, 2, C.byref(got)))
        assert got.should_generate_spectrum == (1 if tick == 0 else 0) and got.tile_length[0] == cascade_preset(2)["tile_length"][0]
        while L.ow_cascades_remaining(gen.context):
            _lib.check(L.ow_process(gen.context))
        _lib.check(L.ow_get_cascade_params(gen.context, 2, C.byref(got)))
        assert got.should_generate_spectrum == 0                                # cleared once processed (wave_generator.gd:72)
        C.memmove(arr, keep, len(keep))
    gen.sync()
    for i in range(len(ids)):
        a, b = gen.get_maps(i), ref.get_maps(i)
        assert np.array_equal(a[0].view(np.uint16), b[0].view(np.uint16)) and np.array_equal(a[1].view(np.uint16), b[1].view(np.uint16))
    assert L.ow_set_cascade_params(gen.context, 3, C.byref(got)) == _lib.OW_ERR_INVALID
    assert L.ow_get_cascade_params(gen.context, -1, C.byref(got)) == _lib.OW_ERR_INVALID


def test_live_edit_between_update_and_process_is_pushed_explicitly():
    """the reference reads the edited parameter object when it processes the cascade; a C caller pushes the edited record"""
    n, ids = 256, [0, 1]
    gen, L = raw_gen(n, 2, debug_f32=True)
    arr = packed(ids)
    _lib.check(L.ow_update(gen.context, UPDATE_DELTA, arr, 2))
    edited = ow_cascade_params()
    _lib.check(L.ow_get_cascade_params(gen.context, 1, C.byref(edited)))
    edited.wind_speed, edited.should_generate_spectrum = 9.0, 1
    _lib.check(L.ow_set_cascade_params(gen.context, 1, C.byref(edited)))
    while L.ow_cascades_remaining(gen.context):
        _lib.check(L.ow_process(gen.context))
    gen.sync()
    og = H.oracle_generator(n, ids)
    og.params[1].wind_speed = 9.0
    og.update_all(UPDATE_DELTA)
    for i in range(2):
        f32, want = gen.get_maps_f32(i), og.f32(i)
        for c, name in enumerate(H.CHANNELS):
            if name != "foam":
                assert H.relmax(f32[..., c], want[..., c]) < H.TOL_F32, (i, name)


def test_bad_record_is_refused_on_the_way_in_and_leaves_no_trace():
    """A record the kernels cannot take is refused by ow_update / ow_update_all / ow_set_cascade_params BEFORE anything changes:
    no time advanced, no dirty flag consumed, nothing armed, nothing launched -- so the corrected array simply goes in again and the
    context is never wedged on a stale armed copy (validation used to happen at enqueue, after the arm)."""
    n = 256
    gen, L = raw_gen(n, 2)
    arr = packed([0, 1])
    _lib.check(L.ow_update_all(gen.context, UPDATE_DELTA, arr, 2))
    gen.sync()
    before = [gen.get_maps(i) for i in range(2)]
    t0 = [a.time for a in arr]
    arr[0].whitecap = float("nan")
    arr[1].should_generate_spectrum = 1
    for call in (L.ow_update_all, L.ow_update):
        assert call(gen.context, UPDATE_DELTA, arr, 2) == _lib.OW_ERR_INVALID and b"non-finite" in L.ow_last_error()
        assert [a.time for a in arr] == t0 and arr[1].should_generate_spectrum == 1      # the caller's records are untouched
        assert L.ow_cascades_remaining(gen.context) == 0                                 # nothing armed
    assert _lib.check(L.ow_run(gen.context, UPDATE_DELTA, arr, 2, 0)) is None
    assert L.ow_run(gen.context, UPDATE_DELTA, arr, 2, 5) == _lib.OW_ERR_INVALID and [a.time for a in arr] == t0
    gen.sync()
    for i in range(2):
        after = gen.get_maps(i)
        assert np.array_equal(after[0].view(np.uint16), before[i][0].view(np.uint16))
    arr[0].tile_length[1] = 0.0
    arr[0].whitecap = 0.5
    assert L.ow_update(gen.context, UPDATE_DELTA, arr, 2) == _lib.OW_ERR_INVALID and b"tile_length" in L.ow_last_error()
    arr[0].tile_length[1] = arr[0].tile_length[0]
    # the corrected array: one tick, time advanced exactly once, and the one-cascade-per-frame drain works
    _lib.check(L.ow_update(gen.context, UPDATE_DELTA, arr, 2))
    assert [a.time for a in arr] == [t + UPDATE_DELTA for t in t0] and L.ow_cascades_remaining(gen.context) == 2
    # a bad live edit is refused as well and leaves the armed copy as it was
    bad, kept = ow_cascade_params(), ow_cascade_params()
    _lib.check(L.ow_get_cascade_params(gen.context, 0, C.byref(bad)))
    bad.wind_direction = float("inf")
    assert L.ow_set_cascade_params(gen.context, 0, C.byref(bad)) == _lib.OW_ERR_INVALID
    _lib.check(L.ow_get_cascade_params(gen.context, 0, C.byref(kept)))
    assert kept.wind_direction == arr[0].wind_direction
    while L.ow_cascades_remaining(gen.context):
        _lib.check(L.ow_process(gen.context))
    gen.sync()
    assert L.ow_update(gen.context, float("inf"), arr, 2) == _lib.OW_ERR_INVALID
    assert L.ow_cascades_remaining(gen.context) == 0


def test_setter_clamps_hold_for_a_caller_without_setters():
    """wind_speed = 0 / fetch_length = 0 through the raw ABI == the clamped values through the mirror's setters"""
    n = 128
    gen, L = raw_gen(n, 2)
    arr = packed([0, 1])
    arr[0].wind_speed, arr[1].fetch_length = 0.0, 0.0
    _lib.check(L.ow_update_all(gen.context, UPDATE_DELTA, arr, 2))
    gen.sync()
    ref = WaveGenerator()
    ref.map_size = n
    ref.init_gpu(2)
    params = [WaveCascadeParameters(**cascade_preset(ci)) for ci in (0, 1)]
    params[0].wind_speed, params[1].fetch_length = 0.0, 0.0
    assert params[0].wind_speed == 0.0001 and params[1].fetch_length == 0.0001
    ref.update_all(UPDATE_DELTA, params)
    ref.sync()
    for i in range(2):
        h0, _ = gen.get_spectrum(i)
        r0, _ = ref.get_spectrum(i)
        assert np.isfinite(h0).all() and np.array_equal(h0.view(np.uint32), r0.view(np.uint32))


def test_create_leaves_the_callers_device_current():
    import torch
    before = torch.cuda.current_device()
    gen, _ = raw_gen(128, 2, device_id=0)
    assert torch.cuda.current_device() == before
    gen.free()
    assert torch.cuda.current_device() == before


def test_a_rendezvous_that_gives_up_is_an_error_not_maps():
    """2048^2: a row is a pair of waves that meet through LDS epoch words with a BOUNDED wait.  The test hook makes the second
    wave of every pair withhold its epoch: the first one's wait gives up, says so in the device status word, and the next
    synchronising call fails with OW_ERR_HIP (once); the following tick is clean again and correct."""
    n, ids = 2048, [1]
    gen, L = raw_gen(n, 2, debug_f32=True)
    params = [WaveCascadeParameters(**cascade_preset(ci)) for ci in ids]
    gen.update_all(UPDATE_DELTA, params)
    gen.sync()
    gen.debug_inject_fault(1)
    gen.update_all(UPDATE_DELTA, params)
    with pytest.raises(_lib.OceanWavesError) as e:
        gen.sync()
    assert e.value.status == _lib.OW_ERR_HIP and "rendezvous" in str(e.value)
    gen.sync()                                  # the status word is reported once ...
    with pytest.raises(_lib.OceanWavesError):   # ... but the failure is sticky for what hands out maps, until they are recomputed
        gen.get_maps(0)
    with pytest.raises(_lib.OceanWavesError):
        gen.get_maps_f32(0)
    gen.debug_inject_fault(1)
    gen.update_all(UPDATE_DELTA, params)
    with pytest.raises(_lib.OceanWavesError):   # every call that hands out maps checks, not only ow_sync
        gen.get_maps(0)
    # foam is recurrent state and the two faulty ticks corrupted it: restore it the documented way, then a clean tick
    og = H.oracle_generator(n, ids)
    for _ in range(3):
        og.update_all(UPDATE_DELTA)
    gen.set_normal_map(0, og.normal(0).view(np.float16))
    gen.update_all(UPDATE_DELTA, params)
    og.update_all(UPDATE_DELTA)
    gen.sync()
    f32, want = gen.get_maps_f32(0), og.f32(0)
    for c, name in enumerate(H.CHANNELS):
        if name == "foam":
            assert np.abs(f32[..., c] - want[..., c]).max() <= H.TOL_FOAM_ABS
        else:
            assert H.relmax(f32[..., c], want[..., c]) < H.TOL_F32, name


def test_a_faulted_layer_stays_refused_until_that_layer_is_recomputed():
    """The reference's schedule enqueues ONE cascade per call (wave_generator.gd:56-63).  After a faulted batch of two cascades, an
    ow_process that recomputes cascade 1 lifts the mark of layer 1 only: layer 0 still holds the faulted batch's bytes and stays
    refused (ow_get_maps, ow_sample_surface over it) until its own cascade has been processed again."""
    n, ids = 2048, [1, 2]
    gen, L = raw_gen(n, 2)
    params = [WaveCascadeParameters(**cascade_preset(ci)) for ci in ids]
    gen.update_all(UPDATE_DELTA, params)
    gen.sync()
    gen.debug_inject_fault(1)
    gen.update_all(UPDATE_DELTA, params)       # both layers recomputed by launches that report a failure
    with pytest.raises(_lib.OceanWavesError):
        gen.sync()
    for layer in (0, 1):
        with pytest.raises(_lib.OceanWavesError):
            gen.get_maps(layer)
    gen.update(UPDATE_DELTA, params)
    gen._process(0.0)                          # highest index first: cascade 1 only
    gen.sync()
    gen.get_maps(1)                            # recomputed: handed out again
    with pytest.raises(_lib.OceanWavesError) as e:
        gen.get_maps(0)                        # not recomputed: still the faulted batch's bytes
    assert e.value.status == _lib.OW_ERR_HIP
    scales = [(1 / 57.0, 1 / 57.0, 1.0, 1.0), (1 / 16.0, 1 / 16.0, 1.0, 1.0)]
    with pytest.raises(_lib.OceanWavesError):
        gen.sample_surface([[1.0, 2.0]], scales)    # the consumer's sums run over layer 0 as well
    gen._process(0.0)                          # cascade 0
    gen.sync()
    gen.get_maps(0)
    gen.sample_surface([[1.0, 2.0]], scales)


@pytest.mark.gpu
@pytest.mark.parametrize("n,count", [(1024, 1), (512, 8), (256, 4)])
def test_lazy_scratch_allocates_the_look_aheads_share_on_first_use_and_changes_nothing(n, count):
    """OW_FLAG_LAZY_SCRATCH: ow_create allocates the scratch of ONE batch; what the look-ahead keeps in flight -- the pair kernel two batches,
    the group kernel a ring of five groups -- is allocated by the first call that speculates.  Same calls, same bits as a context that allocated it up
    front; and the lazy context really did start smaller (free device memory right after creation)."""
    import torch
    from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator
    from godotoceanwaves_amd.presets import UPDATE_DELTA, cascade_preset
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    a = WaveGenerator(); a.map_size, a.lazy_scratch = n, True; a.init_gpu(max(2, count))
    free_lazy = torch.cuda.mem_get_info()[0]
    b = WaveGenerator(); b.map_size = n; b.init_gpu(max(2, count))
    free_eager = torch.cuda.mem_get_info()[0]
    assert (free0 - free_lazy) < (free_lazy - free_eager)          # the eager context holds the look-ahead's scratch as well
    pa = [WaveCascadeParameters(**cascade_preset(i)) for i in range(count)]
    pb = [WaveCascadeParameters(**cascade_preset(i)) for i in range(count)]
    for g, p in ((a, pa), (b, pb)):
        for _ in range(5):
            g.update_all(UPDATE_DELTA, p)                            # arms the look-ahead: the lazy context grows its scratch here
        g.update(UPDATE_DELTA, p)
        for _ in range(count):
            g._process(0.0)
        g.run(UPDATE_DELTA, p, 7)
        g.run(UPDATE_DELTA, p, 7)
        g.sync()
    assert a.lookahead_stats() == b.lookahead_stats() and a.lookahead_stats()[0] > 0
    for i in range(count):
        da, na = a.get_maps(i)
        db, nb = b.get_maps(i)
        assert np.array_equal(da.view(np.uint16), db.view(np.uint16)) and np.array_equal(na.view(np.uint16), nb.view(np.uint16))
    a.free(); b.free()
