"""TWO CHAINS (include/ocean_waves.h, ow_kernels.h; round 6): tick-pair launches of four 1024^2 (eight 512^2) cascades a side go out as two launches of
half the cascades on two streams, each half a chain of its own.  Same kernel, same items: whatever is called, the maps are BITWISE those of a context whose launches stay
whole on the one stream (OW_FLAG_SINGLE_STREAM), and everything the context enqueues or waits for is ordered behind BOTH chains -- on a stream of the
caller's the second chain is joined before the call returns."""
import numpy as np
import pytest

from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator
from godotoceanwaves_amd.presets import UPDATE_DELTA, cascade_preset

pytestmark = pytest.mark.gpu


def make(n, count, single, stream=None, maps=None):
    gen = WaveGenerator()
    gen.map_size, gen.single_stream = n, single
    if stream is not None:
        gen.stream = stream
    if maps is not None:
        gen.external_maps = maps
    gen.init_gpu(max(2, count))
    return gen, [WaveCascadeParameters(**cascade_preset(i)) for i in range(count)]


def same(a, b, count):
    a.sync(); b.sync()
    for i in range(count):
        da, na = a.get_maps(i)
        db, nb = b.get_maps(i)
        assert np.array_equal(da.view(np.uint16), db.view(np.uint16)), i
        assert np.array_equal(na.view(np.uint16), nb.view(np.uint16)), i


@pytest.mark.parametrize("n,count", [(1024, 4), (1024, 8), (512, 8)])
def test_two_chains_leave_the_maps_of_one_stream(n, count):
    a, pa = make(n, count, single=False)
    b, pb = make(n, count, single=True)

    def both(f):
        f(a, pa); f(b, pb)

    both(lambda g, p: g.run(UPDATE_DELTA, p, 7))
    split = a.chain_stats()
    assert split > 0 and b.chain_stats() == 0            # the run's pair launches were split / stayed whole
    assert a.last_kernel_family() == b.last_kernel_family() == "tick_pairs_compact"
    same(a, b, count)
    # tick by tick (the look-ahead's pair launches are split as well), a readback and the reference's schedule in between, runs after runs
    both(lambda g, p: [g.update_all(UPDATE_DELTA, p) for _ in range(6)])
    same(a, b, count)

    def mixed(g, p):
        g.run(UPDATE_DELTA, p, 5); g.get_maps(count - 1); g.run(UPDATE_DELTA, p, 5)
        g.update(UPDATE_DELTA, p)
        for _ in range(count):
            g._process(0.0)
        g.run(UPDATE_DELTA, p, 3); g.run(UPDATE_DELTA, p, 3)
    both(mixed)
    assert a.chain_stats() > split and b.chain_stats() == 0
    same(a, b, count)
    # an edit that regenerates a spectrum of the SECOND chain's half, in the middle of the stream
    def wind(g, p):
        p[count - 1].wind_speed = 13.0
        g.run(UPDATE_DELTA, p, 4)
    both(wind)
    same(a, b, count)
    assert [p.time for p in pa] == [p.time for p in pb]
    a.free(); b.free()


@pytest.mark.parametrize("n,count", [(1024, 2), (1024, 3), (1024, 6), (512, 7), (512, 4), (2048, 1), (256, 4)])
def test_nothing_else_is_split(n, count):
    """a half must still fill the chip and both chains must fit the Infinity Cache: only four 1024^2 or eight 512^2 cascades a side are split"""
    a, pa = make(n, count, single=False)
    a.run(UPDATE_DELTA, pa, 6); a.run(UPDATE_DELTA, pa, 6)
    for _ in range(4):
        a.update_all(UPDATE_DELTA, pa)
    a.sync()
    assert a.chain_stats() == 0
    a.free()


def test_callers_stream_sees_both_chains():
    """ow_config.stream: work the CALLER enqueues on its stream after ow_run / ow_update_all has returned finds every map complete -- the second chain
    (cascades 2 and 3, on the context's own side stream) included -- without any host synchronisation"""
    import torch
    n, count = 1024, 4
    ref, pr = make(n, count, single=True)
    stream = torch.cuda.Stream()
    disp = torch.zeros((count, n, n, 4), dtype=torch.float16, device="cuda")
    norm = torch.zeros_like(disp)
    torch.cuda.synchronize()
    gen, pg = make(n, count, single=False, stream=stream.cuda_stream, maps=(disp.data_ptr(), norm.data_ptr()))
    for step in range(3):
        if step == 1:
            for _ in range(3):
                gen.update_all(UPDATE_DELTA, pg); ref.update_all(UPDATE_DELTA, pr)
        else:
            gen.run(UPDATE_DELTA, pg, 9); ref.run(UPDATE_DELTA, pr, 9)
        with torch.cuda.stream(stream):              # ordered by the caller's stream alone
            d_dev, n_dev = disp.clone(), norm.clone()
        stream.synchronize()
        ref.sync()
        for i in range(count):
            dr, nr = ref.get_maps(i)
            assert np.array_equal(d_dev[i].cpu().numpy().view(np.uint16), dr.view(np.uint16)), (step, i)
            assert np.array_equal(n_dev[i].cpu().numpy().view(np.uint16), nr.view(np.uint16)), (step, i)
    assert gen.chain_stats() > 0
    gen.free(); ref.free()


def test_readbacks_and_destruction_with_both_chains_in_flight():
    """the asynchronous hand-off snapshots in the order of BOTH chains (a layer of the second chain's half included), and a context may be destroyed
    with launches of both chains in flight (ow_destroy drains both streams before it frees their scratch).  (No fault-injection case here: the chains
    exist at 1024^2 only, where a row lives in one wave and no kernel waits for another wave -- the bounded rendezvous is a 2048^2 matter.)"""
    n, count = 1024, 4
    a, pa = make(n, count, single=False)
    b, pb = make(n, count, single=True)
    for g, p in ((a, pa), (b, pb)):
        g.run(UPDATE_DELTA, p, 6)
        g.readback_begin([0, 3])               # one layer of either chain
        g.run(UPDATE_DELTA, p, 4)              # the following ticks overlap the copies
    got = {i: a.readback_wait(i) for i in (0, 3)}
    want = {i: b.readback_wait(i) for i in (0, 3)}
    for i in (0, 3):
        for x, y in zip(got[i], want[i]):
            assert np.array_equal(np.asarray(x).view(np.uint16), np.asarray(y).view(np.uint16)), i
    same(a, b, count)
    assert a.chain_stats() > 0
    a.run(UPDATE_DELTA, pa, 40)                # no synchronisation: both chains are busy when the context goes
    a.free(); b.free()
    c, pc = make(n, count, single=False)       # the device is fine afterwards
    d, pd = make(n, count, single=True)
    c.run(UPDATE_DELTA, pc, 3); d.run(UPDATE_DELTA, pd, 3)
    same(c, d, count)
    c.free(); d.free()


def test_on_a_callers_stream_only_long_runs_are_split():
    """every call on a caller's stream ends with the join of the second chain, which costs more than one split launch gains: there only the launches of an
    ow_run of at least eight ticks are split; on the context's own stream (nothing joins until something synchronises) every such launch is"""
    import torch
    n, count = 1024, 4
    stream = torch.cuda.Stream()
    gen, pg = make(n, count, single=False, stream=stream.cuda_stream)
    own, po = make(n, count, single=False)
    for _ in range(6):
        gen.update_all(UPDATE_DELTA, pg); own.update_all(UPDATE_DELTA, po)     # (the look-ahead's pair launches from the third call on)
    assert gen.chain_stats() == 0 and own.chain_stats() > 0
    gen.run(UPDATE_DELTA, pg, 5); gen.run(UPDATE_DELTA, pg, 7)
    assert gen.chain_stats() == 0
    gen.run(UPDATE_DELTA, pg, 8)
    assert gen.chain_stats() > 0
    own.run(UPDATE_DELTA, po, 5); own.run(UPDATE_DELTA, po, 7); own.run(UPDATE_DELTA, po, 8)
    stream.synchronize()
    same(gen, own, count)       # the same calls, split differently: the same bits
    gen.free(); own.free()
