#!/usr/bin/env python3
"""Same-process A/B of ow_run's merged launches (tick pairs / tick groups) against one launch per pass: both contexts alive at once,
timed regions ALTERNATE (merged, unmerged, merged, ...) so that clock state and box are the same for both.
   scripts/ab_merged.py [N:C ...]   ->  us per tick, median and min..max of `reps` regions each"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA
cases = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(1024, 4), (1024, 2), (1024, 3), (512, 8), (1024, 1), (256, 4)]
K, reps = 2000, 7
for n, c in cases:
    gens = {}
    for merged in (True, False):
        g = WaveGenerator(); g.map_size = n; g.tick_groups = merged; g.init_gpu(max(2, c))
        p = [WaveCascadeParameters(**cascade_preset(i)) for i in range(c)]
        g.run(UPDATE_DELTA, p, 300); g.sync()
        gens[merged] = (g, p)
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:  # clock priming
        for g, p in gens.values():
            g.run(UPDATE_DELTA, p, 100); g.sync()
    samples = {True: [], False: []}
    for r in range(reps):
        for merged in ((True, False) if r % 2 == 0 else (False, True)):
            g, p = gens[merged]
            t0 = time.perf_counter(); g.run(UPDATE_DELTA, p, K); g.sync(); samples[merged].append((time.perf_counter() - t0) / K * 1e6)
    fam = {m: gens[m][0].last_kernel_family() for m in gens}
    fmt = lambda v: f"{statistics.median(v):7.2f} [{min(v):.2f}..{max(v):.2f}]"
    print(f"{n}^2 x {c}   merged ({fam[True]}): {fmt(samples[True])} us   one launch per pass ({fam[False]}): {fmt(samples[False])} us", flush=True)
    for g, _ in gens.values():
        g.free()
