#!/usr/bin/env python3
"""Experiment: the same cascades as ONE context (one stream) against SEVERAL contexts of fewer cascades each on streams of their own (cascades are
independent: no dependency between the streams), their ow_run calls enqueued alternately:  two_ctx.py [map_size = 1024] [cascades = 4] [chunk ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
count = int(sys.argv[2]) if len(sys.argv) > 2 else 4
chunks = [int(x) for x in sys.argv[3:]] or [1, 4, 16]
ticks = max(200, min(2000, int(2000 * (1024 * 1024 * 4) / (n * n * count))))
def ctx(ids):
    g = WaveGenerator(); g.map_size = n; g.init_gpu(max(2, len(ids)))
    return g, [WaveCascadeParameters(**cascade_preset(i)) for i in ids]
one = ctx(list(range(count)))
one[0].run(UPDATE_DELTA, one[1], ticks // 2); one[0].sync()
for rep in range(2):
    t0 = time.perf_counter(); one[0].run(UPDATE_DELTA, one[1], ticks); one[0].sync(); dt = time.perf_counter() - t0
    print(f"{n}^2 x {count}: one context         : {dt/ticks*1e6:8.2f} us per tick ({one[0].last_kernel_family_name() if hasattr(one[0], 'last_kernel_family_name') else ''})", flush=True)
one[0].free()
splits = []
if count % 2 == 0: splits.append([list(range(0, count // 2)), list(range(count // 2, count))])
if count % 4 == 0: splits.append([list(range(i * count // 4, (i + 1) * count // 4)) for i in range(4)])
for split in splits:
    cs = [ctx(ids) for ids in split]
    for chunk in chunks:
        for _ in range(max(1, ticks // 3 // chunk)):
            for g, p in cs: g.run(UPDATE_DELTA, p, chunk)
        for g, p in cs: g.sync()
        for rep in range(2):
            t0 = time.perf_counter()
            for _ in range(ticks // chunk):
                for g, p in cs: g.run(UPDATE_DELTA, p, chunk)
            for g, p in cs: g.sync()
            dt = time.perf_counter() - t0
            print(f"{n}^2 x {count}: {len(cs)} contexts, chunk {chunk:3d}: {dt/(ticks // chunk * chunk)*1e6:8.2f} us per tick", flush=True)
    for g, p in cs: g.free()
