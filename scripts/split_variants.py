#!/usr/bin/env python3
"""timing-only experiment driver: 2048^2 x {1,4} ticks with the library named by OCEAN_WAVES_LIB"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA
import sys as _s
CASES = [tuple(int(v) for v in a.split(':')) for a in _s.argv[1:]] or [(2048, 1), (2048, 4)]
for n, c in CASES:
    gen = WaveGenerator(); gen.map_size = n; gen.init_gpu(max(2, c))
    params = [WaveCascadeParameters(**cascade_preset(i)) for i in range(c)]
    gen.run(UPDATE_DELTA, params, 600); gen.sync()
    t0 = time.perf_counter(); gen.run(UPDATE_DELTA, params, 400); gen.sync(); dt = time.perf_counter() - t0
    gen.timing(True); gen.run(UPDATE_DELTA, params, 100); gen.sync(); p1, p2, launches = gen.timing_read(); gen.timing(False)
    print(f"{os.path.basename(os.environ.get('OCEAN_WAVES_LIB', 'product'))}: {n}^2 x {c}  {dt/400*1e6:8.1f} us/tick   p1 {p1*1e3:6.1f} us  p2 {p2*1e3:6.1f} us per launch", flush=True)
    gen.free()
