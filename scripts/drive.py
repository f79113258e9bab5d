#!/usr/bin/env python3
"""Torch-free driver of the hot path (ctypes only) for rocprofv3 runs: N frames of map_size^2 x cascades."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--map-size", type=int, default=1024)
ap.add_argument("--cascades", type=int, default=4)
ap.add_argument("--frames", type=int, default=50)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--single-stream", action="store_true", help="OW_FLAG_SINGLE_STREAM: every tick-pair launch whole, on the one stream (per-launch counters of the whole launch)")
a = ap.parse_args()
gen = WaveGenerator()
gen.map_size = a.map_size
gen.single_stream = a.single_stream
gen.init_gpu(max(2, a.cascades))
params = [WaveCascadeParameters(**cascade_preset(i)) for i in range(a.cascades)]
gen.run(UPDATE_DELTA, params, a.warmup)
gen.sync()
t0 = time.perf_counter()
gen.run(UPDATE_DELTA, params, a.frames)
gen.sync()
dt = time.perf_counter() - t0
print(f"{a.map_size}^2 x {a.cascades}: {a.frames} frames in {dt*1e3:.2f} ms = {dt/a.frames*1e6:.1f} us/frame, {a.frames*a.cascades/dt:.0f} maps/s")
