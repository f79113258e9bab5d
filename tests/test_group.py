"""Cascades sharded over several devices INSIDE one process: the C-ABI's ow_group_* (include/ocean_waves.h "several devices",
SURVEY.md 8e).  The GPU box has ONE MI355X, so the group is two (or four) shards on device 0 -- with
OW_GROUP_FLAG_FORCE_PEER_PATH every shard goes through the whole remote path: snapshot in stream order, side stream,
hipMemcpyPeerAsync (to self) into the consumer's layer slots.  Cascades are independent (wave_generator.gd:65-85), so what a
shard computes is bit for bit what a lone context with the same cascades computes, and the gathered arrays have to hold exactly
those bytes AS OF the gather_begin call, while later ticks overwrite the live maps."""
import glob
import os

import numpy as np
import pytest

import helpers as H
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, WaveGeneratorGroup, _lib
from godotoceanwaves_amd.presets import UPDATE_DELTA, cascade_preset
from oracle import oracle as O

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def lone(n, ids, ticks, run=True):
    gen = WaveGenerator()
    gen.map_size = n
    gen.init_gpu(max(2, len(ids)))
    params = [WaveCascadeParameters(**cascade_preset(ci)) for ci in ids]
    if run:
        gen.run(UPDATE_DELTA, params, ticks)
    else:
        for _ in range(ticks):
            gen.update_all(UPDATE_DELTA, params)
    gen.sync()
    return gen, params


def bits(a):
    return np.asarray(a).view(np.uint16)


@pytest.mark.parametrize("force_peer", [True, False], ids=["peer_path", "same_device_path"])
@pytest.mark.parametrize("n,shards,per", [(256, 2, 2), (512, 4, 1), (1024, 2, 1)])
def test_gathered_arrays_hold_the_snapshot_of_every_shard(n, shards, per, force_peer):
    ids = list(range(shards * per))
    grp = WaveGeneratorGroup()
    grp.map_size = n
    grp.force_peer_path = force_peer
    grp.init_gpu([0] * shards, per, root=shards - 1)
    assert grp.num_cascades == len(ids)
    with pytest.raises(_lib.OceanWavesError) as e:   # nothing gathered yet
        grp.get_maps(0)
    assert e.value.status == _lib.OW_ERR_STATE
    params = [WaveCascadeParameters(**cascade_preset(ci)) for ci in ids]
    grp.run(UPDATE_DELTA, params, 5)
    grp.gather_begin()                 # snapshot of tick 5 ...
    grp.run(UPDATE_DELTA, params, 4)   # ... while four more ticks overwrite the live maps
    with pytest.raises(_lib.OceanWavesError) as e:   # the arrays are being written: no reading them before the wait
        grp.get_maps(0)
    assert e.value.status == _lib.OW_ERR_STATE
    grp.gather_wait()
    ms, nbytes = grp.gather_stats()
    assert nbytes == per * n * n * 16 and ms > 0.0
    for s in range(shards):
        twin, _ = lone(n, ids[s * per:(s + 1) * per], 5)
        for l in range(per):
            want_d, want_n = twin.get_maps(l)
            got_d, got_n = grp.get_maps(s * per + l)
            assert np.array_equal(bits(got_d), bits(want_d)) and np.array_equal(bits(got_n), bits(want_n)), (s, l)
            live_d, _ = grp.shard(s).get_maps(l)   # the live maps have moved on
            assert not np.array_equal(bits(live_d), bits(want_d))
        twin.free()
    # a second gather replaces the first (the snapshot buffer is reused only after its copy has left)
    grp.gather_begin()
    grp.gather_wait()
    grp.sync()
    for s in range(shards):
        for l in range(per):
            live = grp.shard(s).get_maps(l)
            got = grp.get_maps(s * per + l)
            assert np.array_equal(bits(got[0]), bits(live[0])) and np.array_equal(bits(got[1]), bits(live[1]))
    assert params[0].time == pytest.approx(120.0 + 9 * UPDATE_DELTA)
    grp.free()


def test_group_keeps_the_reference_schedule_one_cascade_per_frame_highest_index_first():
    """ow_group_update arms every shard; ow_group_process drains one cascade per call, highest GLOBAL index first
    (wave_generator.gd:56-63); the result equals update_all on the same records."""
    n, shards, per = 256, 2, 2
    ids = list(range(shards * per))
    a, b = WaveGeneratorGroup(), WaveGeneratorGroup()
    for g in (a, b):
        g.map_size = n
        g.init_gpu([0] * shards, per)
    pa = [WaveCascadeParameters(**cascade_preset(ci)) for ci in ids]
    pb = [WaveCascadeParameters(**cascade_preset(ci)) for ci in ids]
    for tick in range(3):
        a.update(UPDATE_DELTA, pa)
        assert a.pass_num_cascades_remaining == 4
        order = []
        while a.pass_num_cascades_remaining:
            rem = [int(a._lib.ow_cascades_remaining(a._lib.ow_group_context(a.group, s))) for s in range(shards)]
            order.append(rem)
            a._process()
        assert order == [[2, 2], [2, 1], [2, 0], [1, 0]]
        a._process()  # nothing armed: a no-op
        b.update_all(UPDATE_DELTA, pb)
    # leftovers are flushed by the next update (wave_generator.gd:94-98): arm, drain one, update again
    a.update(UPDATE_DELTA, pa)
    a._process()
    a.update(UPDATE_DELTA, pa)
    while a.pass_num_cascades_remaining:
        a._process()
    b.update_all(UPDATE_DELTA, pb)
    b.update_all(UPDATE_DELTA, pb)
    for g in (a, b):
        g.gather_begin()
        g.gather_wait()
    for c in ids:
        da, na = a.get_maps(c)
        db, nb = b.get_maps(c)
        assert np.array_equal(bits(da), bits(db)) and np.array_equal(bits(na), bits(nb)), c
    assert [p.time for p in pa] == [p.time for p in pb]
    a.free()
    b.free()


def test_c4_shape_against_the_reference_shaders_bytes():
    """BASELINE config C4 in miniature -- 1024^2, ONE cascade per shard, gather to the consumer: after ow_group_run(3) the gathered
    layer of every global cascade that has a committed fixture is held to the bytes the reference's own shaders produced
    (tests/golden/ref_n1024_c*_f3.npz)."""
    fixtures = {int(np.load(p)["cascade"]): p for p in sorted(glob.glob(os.path.join(GOLDEN, "ref_n1024_c*_f3.npz")))}
    assert fixtures
    shards = max(fixtures) + 1
    grp = WaveGeneratorGroup()
    grp.map_size = 1024
    grp.force_peer_path = True
    grp.init_gpu([0] * shards, 1)
    params = [WaveCascadeParameters(**cascade_preset(ci)) for ci in range(shards)]
    grp.run(UPDATE_DELTA, params, 3)
    grp.gather_begin()
    grp.gather_wait()
    assert grp.shard(0).last_kernel_family() == "tick_groups_compact"
    for ci, path in fixtures.items():
        z = np.load(path)
        stride = int(z["row_stride"])
        disp, norm = grp.get_maps(ci)
        assert H.fp16_close(disp[::stride], z["displacement"]) <= 1.0
        assert H.fp16_close(norm[::stride][..., :3], z["normal"][..., :3]) <= 1.0
        foam, foam_ref = norm[::stride][..., 3].astype(np.float64), z["normal"][..., 3].view(np.float16).astype(np.float64)
        assert np.abs(foam - foam_ref).max() <= H.TOL_FOAM_ABS
    grp.free()


def test_consumer_sampling_over_the_gathered_arrays_is_the_oracles():
    """what the consumer's shaders see on the root device (water.gdshader:31-37,72-82 over all cascades) = the oracle's restatement
    evaluated on the gathered bytes, bit for bit"""
    n, shards, per = 256, 3, 1
    grp = WaveGeneratorGroup()
    grp.map_size = n
    grp.force_peer_path = True
    grp.init_gpu([0] * shards, per, root=1)
    params = [WaveCascadeParameters(**cascade_preset(ci)) for ci in range(3)]
    grp.run(UPDATE_DELTA, params, 4)
    grp.gather_begin()
    grp.gather_wait()
    maps = [grp.get_maps(c) for c in range(3)]
    d, m = np.stack([x[0] for x in maps]), np.stack([x[1] for x in maps])
    scales = np.array([[1 / p.tile_length[0], 1 / p.tile_length[1], p.displacement_scale, p.normal_scale] for p in params], np.float32)
    rng = np.random.default_rng(3)
    xz = rng.uniform(-300, 300, (2000, 2)).astype(np.float32)
    got, want = grp.sample_surface(xz, scales), O.sample_surface(d, m, scales, xz)
    for f in ("displacement", "gradient", "gradient_scaled", "foam", "spray_active", "gradient_fragment", "foam_fragment"):
        assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), f
    grp.free()


def test_group_argument_errors_and_bad_records():
    L = _lib.load()
    import ctypes as C
    cfg = _lib.ow_group_config(map_size=256, num_devices=2, cascades_per_device=1, root=2)
    g = C.c_void_p()
    assert L.ow_group_create(C.byref(cfg), C.byref(g)) == _lib.OW_ERR_INVALID and not g
    cfg.root = 0
    cfg.device_ids[1] = 99
    assert L.ow_group_create(C.byref(cfg), C.byref(g)) == _lib.OW_ERR_INVALID and b"device_ids[1]" in L.ow_last_error()
    cfg.device_ids[1] = 0
    cfg.map_size = 300
    assert L.ow_group_create(C.byref(cfg), C.byref(g)) == _lib.OW_ERR_INVALID
    grp = WaveGeneratorGroup()
    grp.map_size = 256
    grp.init_gpu([0, 0], 1)
    with pytest.raises(_lib.OceanWavesError) as e:
        grp.gather_wait()
    assert e.value.status == _lib.OW_ERR_STATE
    params = [WaveCascadeParameters(**cascade_preset(ci)) for ci in range(2)]
    with pytest.raises(_lib.OceanWavesError) as e:   # all cascades' records, or none
        grp.update_all(UPDATE_DELTA, params[:1])
    assert e.value.status == _lib.OW_ERR_INVALID
    grp.update_all(UPDATE_DELTA, params)
    params[1].whitecap = float("nan")
    t0 = [p.time for p in params]
    with pytest.raises(_lib.OceanWavesError) as e:   # all shards' records are checked before any shard starts: nobody is a tick ahead
        grp.update_all(UPDATE_DELTA, params)
    assert e.value.status == _lib.OW_ERR_INVALID and "cascade 1" in str(e.value) and [p.time for p in params] == t0
    grp.free()


def test_resharding_through_the_checkpoint_state():
    """SURVEY.md section 5 (checkpoint / resume): the persistent state of a cascade is its `time`, its parameters + seed (the spectrum is
    regenerated from them) and the FP16 foam channel.  A run on 2 shards x 2 cascades is stopped, its state carried over -- the gathered
    normal maps through ow_set_normal_map, the parameter objects as they are with their dirty flags raised -- into 4 shards x 1 cascade,
    and continued: bit for bit the maps of the run that was never interrupted."""
    n, ids = 256, [0, 1, 2, 3]

    def group(shards, per):
        g = WaveGeneratorGroup()
        g.map_size = n
        g.force_peer_path = True
        g.init_gpu([0] * shards, per)
        return g

    def gathered(g):
        g.gather_begin()
        g.gather_wait()
        return [g.get_maps(c) for c in ids]

    ref = group(2, 2)
    pr = [WaveCascadeParameters(**cascade_preset(ci)) for ci in ids]
    ref.run(UPDATE_DELTA, pr, 6)
    ref.run(UPDATE_DELTA, pr, 5)
    want = gathered(ref)
    ref.free()

    a = group(2, 2)
    pa = [WaveCascadeParameters(**cascade_preset(ci)) for ci in ids]
    a.run(UPDATE_DELTA, pa, 6)
    state = gathered(a)          # the checkpoint: maps (foam in normal.a) + the parameter objects (time inside)
    a.free()

    b = group(4, 1)              # another partition of the same cascades
    for c in ids:
        b.shard(c).set_normal_map(0, state[c][1])
        pa[c].should_generate_spectrum = True    # the new owner has no spectrum yet: regenerated from parameters + seed
    b.run(UPDATE_DELTA, pa, 5)
    got = gathered(b)
    for c in ids:
        assert np.array_equal(bits(got[c][0]), bits(want[c][0])) and np.array_equal(bits(got[c][1]), bits(want[c][1])), c
    assert [p.time for p in pa] == [p.time for p in pr]
    b.free()


def test_a_faulted_shard_keeps_its_gathered_layers_refused_until_a_clean_gather():
    """ow_group_gather_wait reports a shard's device-side failure ONCE -- but the garbage it copied stays in the gathered arrays, so the
    group's readers (ow_group_get_maps of that shard's layers, ow_group_sample_surface over them) keep refusing until a later gather of
    that shard has landed cleanly; the other shard's layers are handed out all along; every pending shard is waited for."""
    n, shards, per = 2048, 2, 1
    grp = WaveGeneratorGroup()
    grp.map_size, grp.force_peer_path = n, True
    grp.init_gpu([0] * shards, per)
    params = [WaveCascadeParameters(**cascade_preset(ci)) for ci in range(shards * per)]
    grp.update_all(UPDATE_DELTA, params)
    grp.gather_begin()
    grp.gather_wait()
    grp.get_maps(0), grp.get_maps(1)
    grp.shard(1).debug_inject_fault(1)                 # shard 1's next batch: a wave-pair rendezvous gives up (2048^2 kernels)
    grp.update_all(UPDATE_DELTA, params)
    grp.gather_begin()
    with pytest.raises(_lib.OceanWavesError) as e:
        grp.gather_wait()
    assert e.value.status == _lib.OW_ERR_HIP and "shard 1" in str(e.value)
    with pytest.raises(_lib.OceanWavesError) as e:
        grp.gather_wait()                              # nothing is left pending: every shard was waited for
    assert e.value.status == _lib.OW_ERR_STATE
    clean, _ = lone(n, [0], 2, run=False)
    d0, m0 = grp.get_maps(0)                           # shard 0's layer: fine, and the right bytes
    assert np.array_equal(bits(d0), bits(clean.get_maps(0)[0])) and np.array_equal(bits(m0), bits(clean.get_maps(0)[1]))
    scales = [(1 / 88.0, 1 / 88.0, 1.0, 1.0), (1 / 57.0, 1 / 57.0, 1.0, 1.0)]
    for _ in range(2):                                 # sticky: not consumed by the first refusal
        with pytest.raises(_lib.OceanWavesError) as e:
            grp.get_maps(1)
        assert e.value.status == _lib.OW_ERR_HIP
        with pytest.raises(_lib.OceanWavesError):
            grp.sample_surface([[3.0, 4.0]], scales)
    grp.sample_surface([[3.0, 4.0]], scales[:1])       # sums over layer 0 only: allowed
    grp.update_all(UPDATE_DELTA, params)               # the shard recomputes its layer ...
    grp.gather_begin()
    grp.gather_wait()                                  # ... and a gather of it lands cleanly: handed out again
    grp.get_maps(1)
    grp.sample_surface([[3.0, 4.0]], scales)
    grp.free()


def test_c4_exact_shape_eight_shards_overlapped_gathers_against_the_oracle():
    """BASELINE config C4's exact shape on the one device of the box: 1024^2, EIGHT shards of one cascade each, every shard through the whole
    remote path (OW_GROUP_FLAG_FORCE_PEER_PATH: snapshot in stream order, side stream, hipMemcpyPeerAsync into the consumer's layer slot),
    210 ticks through ow_group_run with a gather begun every 16 ticks and left to overlap the next chunk -- then all eight gathered layers are
    held to the ORACLE after the same 210 ticks (FP16 maps within one ulp + 1e-5 of the channel maximum, the recurrent foam within one
    step), and to the live maps of their shards bit for bit.  (the group tests used at most four shards.)"""
    n, shards, ticks, every = 1024, 8, 210, 16
    grp = WaveGeneratorGroup()
    grp.map_size = n
    grp.force_peer_path = True
    grp.init_gpu([0] * shards, 1, root=0)
    assert grp.num_cascades == 8
    params = [WaveCascadeParameters(**cascade_preset(ci)) for ci in range(shards)]
    done = 0
    while done < ticks:
        k = min(every, ticks - done)
        grp.run(UPDATE_DELTA, params, k)
        grp.gather_begin()            # overlaps the next chunk's ticks; a new gather waits for the previous one's snapshot buffer only
        done += k
    grp.gather_wait()
    grp.sync()
    ms, nbytes = grp.gather_stats()
    assert nbytes == n * n * 16 and ms > 0.0
    og = H.oracle_generator(n, list(range(shards)), native=True)
    for _ in range(ticks):
        og.update_all(UPDATE_DELTA)
    for c in range(shards):
        got_d, got_n = grp.get_maps(c)
        live_d, live_n = grp.shard(c).get_maps(0)
        assert np.array_equal(bits(got_d), bits(live_d)) and np.array_equal(bits(got_n), bits(live_n)), c   # the last gather followed the last tick
        assert H.fp16_close(got_d, og.displacement(c)) <= 1.0, c
        assert H.fp16_close(got_n[..., :3], og.normal(c)[..., :3]) <= 1.0, c
        foam_err = np.abs(got_n[..., 3].astype(np.float64) - og.normal(c)[..., 3].view(np.float16).astype(np.float64))
        # (recurrent FP16 state over 210 ticks, as in tests/test_golden.py's 1000-frame loop: within two steps everywhere, within one on all but a vanishing share)
        assert foam_err.max() <= 2 * H.TOL_FOAM_ABS and (foam_err > H.TOL_FOAM_ABS).mean() < 1e-3, (c, foam_err.max())
    assert params[7].time == pytest.approx(120.0 + np.pi * 7 + ticks * UPDATE_DELTA)
    og.close()
    grp.free()


@pytest.mark.parametrize("force_peer", [True, False], ids=["peer_path", "same_device_path"])
def test_link_info_says_how_each_shard_reaches_the_root(force_peer):
    """ow_group_link_info / ow_query_link: what the HIP runtime reports between a shard's device and the root's, so
    that the first gather measured on a real node can be read against the right model.  On the one-GPU box every shard sits on the root's
    device: same_device, zero hops, and the path is the staged one exactly when OW_GROUP_FLAG_FORCE_PEER_PATH asks for it."""
    import ctypes as C
    grp = WaveGeneratorGroup()
    grp.map_size = 256
    grp.force_peer_path = force_peer
    grp.init_gpu([0, 0, 0], 1)
    links = grp.link_info()
    assert len(links) == 3
    for l in links:
        assert l["device"] == 0 and l["root_device"] == 0 and l["same_device"] and l["peer_access"] and l["hops"] == 0
        assert l["staged_path"] == force_peer
    lib = _lib.load()
    lk = _lib.ow_group_link()
    assert lib.ow_query_link(0, 0, C.byref(lk)) == _lib.OW_OK and lk.same_device == 1
    assert lib.ow_query_link(0, 99, C.byref(lk)) == _lib.OW_ERR_INVALID      # no such device
    assert lib.ow_group_link_info(grp.group, 7, C.byref(lk)) == _lib.OW_ERR_INVALID
    grp.free()
