#!/usr/bin/env python3
"""where do tick pairs on the layer-parallel compact family beat the runtime's default choice?  us per tick of ow_run"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA
cases = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(1024, 2), (1024, 3), (512, 5), (512, 6), (512, 8), (256, 8), (1024, 1)]
for n, c in cases:
    row = []
    for mode, pairs in ((None, True), ("layer_parallel_compact", True), ("layer_parallel_compact", False), ("compact", True)):
        gen = WaveGenerator(); gen.map_size = n; gen.kernels = mode; gen.tick_groups = pairs; gen.init_gpu(max(2, c))
        params = [WaveCascadeParameters(**cascade_preset(i)) for i in range(c)]
        gen.run(UPDATE_DELTA, params, 1500); gen.sync()
        t0 = time.perf_counter(); gen.run(UPDATE_DELTA, params, 1000); gen.sync(); dt = time.perf_counter() - t0
        row.append(f"{str(mode)[:8]}{'+pairs' if pairs else ''}: {dt/1000*1e6:6.1f} ({gen.last_kernel_family()[:10]})")
        gen.free()
    print(f"{n}^2 x {c}  " + "   ".join(row), flush=True)
