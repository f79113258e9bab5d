#!/usr/bin/env python3
"""Torch-free driver of the reference scene's cadence for rocprofv3 runs (bench.scene_frames: water.gd:75-82's rate limiter over 144 / 60 Hz frames
with 5 % jitter, one ow_process per frame, leftovers flushed by the next ow_update): prints us of GPU per update and the launches a kernel trace
of this process should show per update.     python scripts/scene_probe.py [--map-size 1024] [--cascades 4] [--hz 144] [--frames 1350]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset  # noqa: E402
from bench import Driver, scene_frames  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--map-size", type=int, default=1024)
ap.add_argument("--cascades", type=int, default=4)
ap.add_argument("--hz", type=int, default=144)
ap.add_argument("--frames", type=int, default=1350)
ap.add_argument("--jitter", type=float, default=0.05)
a = ap.parse_args()
gen = WaveGenerator()
gen.map_size = a.map_size
gen.init_gpu(max(2, a.cascades))
drv = Driver(gen, [WaveCascadeParameters(**cascade_preset(i)) for i in range(a.cascades)])
scene_frames(drv, a.hz, 300, a.jitter)
drv.sync()
h0, s0 = gen.lookahead_stats()
t0 = time.perf_counter()
updates = scene_frames(drv, a.hz, a.frames, a.jitter, seed=777)
drv.sync()
dt = time.perf_counter() - t0
h1, s1 = gen.lookahead_stats()
print(f"{a.map_size}^2 x {a.cascades} at {a.hz} Hz (jitter {a.jitter}): {updates} updates in {a.frames} frames, {dt / updates * 1e6:.2f} us per update; "
      f"served from work computed ahead {h1 - h0} (= {(h1 - h0) / max(1, updates * a.cascades):.3f} per cascade-update), launches that carried work ahead {s1 - s0} "
      f"(= {(s1 - s0) / max(1, updates):.3f} per update)")
