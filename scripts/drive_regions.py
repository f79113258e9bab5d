#!/usr/bin/env python3
"""Torch-free driver for a rocprofv3 kernel trace of SHORT regions: `regions` times [ow_run(K ticks), synchronise], as bench.py's driver-sized
regions do (--steps 20).  scripts/trace_regions.py turns the trace into the launch sequence of one region: kernel, duration, gap before it.
    rocprofv3 --kernel-trace --output-format csv -d out -- python scripts/drive_regions.py --map-size 1024 --cascades 8 --ticks 20"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA
ap = argparse.ArgumentParser()
ap.add_argument("--map-size", type=int, default=1024)
ap.add_argument("--cascades", type=int, default=4)
ap.add_argument("--ticks", type=int, default=20)
ap.add_argument("--regions", type=int, default=150)
ap.add_argument("--mode", choices=("run", "calls", "unmerged"), default="run")
a = ap.parse_args()
g = WaveGenerator(); g.map_size = a.map_size
g.run_as_calls, g.tick_groups = a.mode == "calls", a.mode != "unmerged"
g.init_gpu(max(2, a.cascades))
p = [WaveCascadeParameters(**cascade_preset(i)) for i in range(a.cascades)]
g.run(UPDATE_DELTA, p, 400); g.sync()
t0 = time.perf_counter()
for _ in range(a.regions):
    g.run(UPDATE_DELTA, p, a.ticks); g.sync()
dt = time.perf_counter() - t0
print(f"{a.map_size}^2 x {a.cascades} {a.mode}: {a.regions} regions of {a.ticks} ticks, {dt / a.regions / a.ticks * 1e6:.2f} us per tick (host clock, Python mirror's packing included)")
