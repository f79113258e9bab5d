"""Asynchronous hand-off of finished layers to host memory: ow_readback_begin / ow_readback_wait
deliver exactly the bytes ow_get_maps would have returned at the moment of the call, while later ticks overlap the
copy."""
import numpy as np
import pytest

from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, _lib
from godotoceanwaves_amd.presets import UPDATE_DELTA, cascade_preset

pytestmark = pytest.mark.gpu


def make(n, ids):
    gen = WaveGenerator()
    gen.map_size = n
    gen.init_gpu(len(ids))
    return gen, [WaveCascadeParameters(**cascade_preset(ci)) for ci in ids]


def test_readback_is_a_snapshot_in_stream_order():
    gen, params = make(256, [0, 1, 2])
    gen.run(UPDATE_DELTA, params, 3)
    before = [gen.get_maps(i) for i in range(3)]
    gen.readback_begin([0, 2])
    gen.run(UPDATE_DELTA, params, 5)                 # overlaps the copy and overwrites the live maps
    for i in (2, 0):
        d, m = gen.readback_wait(i)
        assert np.array_equal(d.view(np.uint16), before[i][0].view(np.uint16))
        assert np.array_equal(m.view(np.uint16), before[i][1].view(np.uint16))
    after = gen.get_maps(0)
    assert not np.array_equal(after[0].view(np.uint16), before[0][0].view(np.uint16))
    # a second round reuses the staging memory and now carries the newer state
    gen.readback_begin([0])
    d, m = gen.readback_wait(0)
    assert np.array_equal(d.view(np.uint16), after[0].view(np.uint16)) and np.array_equal(m.view(np.uint16), after[1].view(np.uint16))


def test_back_to_back_readbacks_of_one_layer_do_not_tear():
    """begin / begin / wait: the second snapshot must wait for the first PCIe copy instead of overwriting its source"""
    gen, params = make(512, [0])
    gen.run(UPDATE_DELTA, params, 2)
    for _ in range(6):
        gen.readback_begin([0])
        gen.update_all(UPDATE_DELTA, params)
    last = gen.get_maps(0)
    gen.readback_begin([0])
    d, m = gen.readback_wait(0)
    assert np.array_equal(d.view(np.uint16), last[0].view(np.uint16)) and np.array_equal(m.view(np.uint16), last[1].view(np.uint16))


def test_one_cascade_per_frame_drain_with_readback():
    """the reference's schedule (wave_generator.gd:56-63): one cascade per rendered frame, each handed off as it finishes"""
    gen, params = make(128, [0, 1, 2, 3])
    gen.update_all(UPDATE_DELTA, params)
    gen.update(UPDATE_DELTA, params)
    order = []
    while gen.pass_num_cascades_remaining:
        idx = gen.pass_num_cascades_remaining - 1
        gen._process()
        gen.readback_begin([idx])
        order.append(idx)
    assert order == [3, 2, 1, 0]
    for idx in order:
        d, m = gen.readback_wait(idx)
        live = gen.get_maps(idx)
        assert np.array_equal(d.view(np.uint16), live[0].view(np.uint16)) and np.array_equal(m.view(np.uint16), live[1].view(np.uint16))


def test_readback_errors():
    gen, params = make(128, [0, 1])
    gen.update_all(UPDATE_DELTA, params)
    with pytest.raises(_lib.OceanWavesError) as e:
        gen.readback_wait(0)                          # nothing outstanding
    assert e.value.status == _lib.OW_ERR_STATE
    with pytest.raises(_lib.OceanWavesError) as e:
        gen.readback_begin([5])                       # layer outside the context
    assert e.value.status == _lib.OW_ERR_INVALID
    with pytest.raises(_lib.OceanWavesError) as e:
        gen.readback_begin([])
    assert e.value.status == _lib.OW_ERR_INVALID
    gen.readback_begin([1])
    gen.readback_wait(1)
    with pytest.raises(_lib.OceanWavesError) as e:
        gen.readback_wait(1)                          # already consumed
    assert e.value.status == _lib.OW_ERR_STATE
    gen.readback_begin([0, 1])
    gen.free()                                        # destroying with a copy in flight must be clean
