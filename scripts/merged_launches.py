#!/usr/bin/env python3
"""ow_run with its launches merged across ticks (tick groups / tick pairs) against one launch per pass: us per tick, and whether the
maps after 37 ticks are bit-identical.   scripts/merged_launches.py [N:C ...]"""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA
cases = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(256, 4), (512, 4), (512, 8), (1024, 1), (1024, 2), (1024, 3), (1024, 4), (1024, 5)]
for n, c in cases:
    row, sums = [], []
    for merged in (True, False):
        best = 1e9
        gen = WaveGenerator(); gen.map_size = n; gen.tick_groups = merged; gen.init_gpu(max(2, c))
        params = [WaveCascadeParameters(**cascade_preset(i)) for i in range(c)]
        gen.run(UPDATE_DELTA, params, 37); gen.sync()
        h = hashlib.sha1()
        for i in range(c):
            d, m = gen.get_maps(i)
            h.update(d.tobytes() + m.tobytes())
        sums.append(h.hexdigest()[:12])
        gen.run(UPDATE_DELTA, params, 500); gen.sync()
        for rep in range(5):
            t0 = time.perf_counter(); gen.run(UPDATE_DELTA, params, 1000); gen.sync(); best = min(best, time.perf_counter() - t0)
        row.append(f"{best/1000*1e6:7.2f} us ({gen.last_kernel_family()}, depth {gen.tick_group_depth()})")
        gen.free()
    print(f"{n}^2 x {c}  " + "   ".join(row) + ("   identical" if len(set(sums)) == 1 else f"   DIFFERENT {sums}"), flush=True)
