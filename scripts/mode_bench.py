#!/usr/bin/env python3
"""Times ticks of map_size^2 x cascades with a pinned kernel family (standard / layer_parallel / auto)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA
for n, c in [(2048, 1), (2048, 4), (1024, 1), (1024, 2), (1024, 3), (512, 4), (512, 8)]:
    for mode in (None, "standard", "layer_parallel"):
        gen = WaveGenerator(); gen.map_size = n; gen.kernels = mode; gen.init_gpu(max(2, c))
        params = [WaveCascadeParameters(**cascade_preset(i)) for i in range(c)]
        gen.run(UPDATE_DELTA, params, 300); gen.sync()
        t0 = time.perf_counter(); gen.run(UPDATE_DELTA, params, 600); gen.sync(); dt = time.perf_counter() - t0
        print(f"{n}^2 x {c} {str(mode):15s} {dt/600*1e6:8.1f} us/tick", flush=True)
        gen.free()
