#!/usr/bin/env python3
"""Two chains against one (OW_FLAG_SINGLE_STREAM): same calls on two contexts, maps compared bit for bit, then timed alternately.
   chains_ab.py [map_size = 1024] [cascades = 4] [ticks per region = 20 200 2000]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
count = int(sys.argv[2]) if len(sys.argv) > 2 else 4
regions = [int(x) for x in sys.argv[3:]] or [20, 200, 2000]
def ctx(single):
    g = WaveGenerator(); g.map_size = n; g.single_stream = single; g.init_gpu(max(2, count))
    return g, [WaveCascadeParameters(**cascade_preset(i)) for i in range(count)]
a, pa = ctx(False)
b, pb = ctx(True)
def same(tag):
    a.sync(); b.sync()
    for i in range(count):
        da, na = a.get_maps(i); db, nb = b.get_maps(i)
        assert np.array_equal(da.view(np.uint16), db.view(np.uint16)) and np.array_equal(na.view(np.uint16), nb.view(np.uint16)), (tag, i)
    print(f"  {tag}: maps bit-identical; launches split so far: {a.chain_stats()} / {b.chain_stats()}", flush=True)
for g, p in ((a, pa), (b, pb)):
    g.run(UPDATE_DELTA, p, 7)
same("run of 7")
for g, p in ((a, pa), (b, pb)):
    for _ in range(6): g.update_all(UPDATE_DELTA, p)
same("six update_all calls")
for g, p in ((a, pa), (b, pb)):
    g.run(UPDATE_DELTA, p, 5); g.get_maps(0); g.run(UPDATE_DELTA, p, 5); g.update(UPDATE_DELTA, p)
    for _ in range(count): g._process(0.0)
    g.run(UPDATE_DELTA, p, 3); g.run(UPDATE_DELTA, p, 3)
same("runs, a readback in between, the reference's schedule, runs")
for K in regions:
    reps = max(3, 4000 // K)
    for g, p in ((a, pa), (b, pb)):
        g.run(UPDATE_DELTA, p, K); g.sync()
    res = {}
    for rnd in range(3):
        for name, g, p in (("two chains", a, pa), ("one stream", b, pb)):
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter(); g.run(UPDATE_DELTA, p, K); g.sync(); ts.append((time.perf_counter() - t0) / K * 1e6)
            res.setdefault(name, []).append(float(np.median(ts)))
    print(f"{n}^2 x {count}, regions of {K} ticks (one ow_run + sync each), median us per tick, three rounds: " + "; ".join(f"{k}: " + " ".join(f"{v:.2f}" for v in vs) for k, vs in res.items()), flush=True)
same("after the timed regions")
