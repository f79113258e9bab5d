#!/usr/bin/env python3
"""Experiment: the same 4 cascades of 1024^2 as ONE context (one stream, 4 slots per launch) vs TWO contexts of 2 cascades on
their own streams (independent cascades: no dependency between the streams), ticks enqueued alternately."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
def ctx(ids):
    g = WaveGenerator(); g.map_size = n; g.init_gpu(max(2, len(ids)))
    return g, [WaveCascadeParameters(**cascade_preset(i)) for i in ids]
one = ctx([0, 1, 2, 3])
one[0].run(UPDATE_DELTA, one[1], 1500); one[0].sync()
t0 = time.perf_counter(); one[0].run(UPDATE_DELTA, one[1], 2000); one[0].sync(); dt = time.perf_counter() - t0
print(f"one context x4       : {dt/2000*1e6:7.1f} us per tick of 4 cascades")
for split in ([[0, 1], [2, 3]], [[0], [1], [2], [3]]):
    cs = [ctx(ids) for ids in split]
    for chunk in (1, 4, 16):
        for _ in range(600 // chunk):
            for g, p in cs: g.run(UPDATE_DELTA, p, chunk)
        for g, p in cs: g.sync()
        t0 = time.perf_counter()
        for _ in range(2000 // chunk):
            for g, p in cs: g.run(UPDATE_DELTA, p, chunk)
        for g, p in cs: g.sync()
        dt = time.perf_counter() - t0
        print(f"{len(cs)} contexts, chunk {chunk:2d} : {dt/2000*1e6:7.1f} us per tick of 4 cascades")
    for g, p in cs: g.free()
