#!/usr/bin/env python3
"""Times ticks of map_size^2 x cascades with a pinned kernel family (standard / layer_parallel / auto), with the
in-situ per-kernel durations.  Usage: python scripts/mode_bench.py [n:c ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA
cases = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(2048, 1), (2048, 4), (1024, 1), (1024, 2), (1024, 4), (512, 1), (512, 4), (512, 8), (256, 1), (256, 4), (128, 4)]
for n, c in cases:
    for mode in (None, "standard", "layer_parallel", "compact", "layer_parallel_compact"):
        gen = WaveGenerator(); gen.map_size = n; gen.kernels = mode; gen.init_gpu(max(2, c))
        params = [WaveCascadeParameters(**cascade_preset(i)) for i in range(c)]
        gen.run(UPDATE_DELTA, params, 1500); gen.sync()
        t0 = time.perf_counter(); gen.run(UPDATE_DELTA, params, 1000); gen.sync(); dt = time.perf_counter() - t0
        gen.timing(True); gen.run(UPDATE_DELTA, params, 200); gen.sync(); p1, p2, launches = gen.timing_read(); gen.timing(False)
        print(f"{n}^2 x {c} {str(mode):15s} {dt/1000*1e6:8.1f} us/tick   p1 {p1*1e3:6.1f} us  p2 {p2*1e3:6.1f} us per launch ({launches // 200} launch pairs/tick)", flush=True)
        gen.free()
