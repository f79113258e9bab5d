#!/usr/bin/env python3
"""PCIe-inclusive rates of the host hand-off: what a consumer that needs the maps in host
memory (RenderingDevice.texture_update) gets, next to the in-HBM rate bench.py reports.
  a) every tick of 1024^2 x 4, all four layers read back, pipelined (readback of tick k overlaps tick k+1)
  b) the reference's schedule: one cascade per rendered frame, that layer read back each frame
  c) synchronous ow_get_maps into pageable memory (debug path)
Prints one JSON object.  Usage: python scripts/pcie_rate.py [--map-size 1024] [--cascades 4] [--ticks 200]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA

ap = argparse.ArgumentParser()
ap.add_argument("--map-size", type=int, default=1024)
ap.add_argument("--cascades", type=int, default=4)
ap.add_argument("--ticks", type=int, default=200)
a = ap.parse_args()
n, C, K = a.map_size, a.cascades, a.ticks
gen = WaveGenerator(); gen.map_size = n; gen.init_gpu(C)
params = [WaveCascadeParameters(**cascade_preset(i)) for i in range(C)]
gen.run(UPDATE_DELTA, params, 50); gen.sync()
layer_mb = n * n * 16 / 1e6
out = {"map_size": n, "cascades": C, "bytes_per_map": n * n * 16}

t0 = time.perf_counter(); gen.run(UPDATE_DELTA, params, K); gen.sync(); dt = time.perf_counter() - t0
out["in_hbm_maps_per_s"] = round(K * C / dt, 1)

# a) pipelined full readback
gen.readback_begin(range(C)); [gen.readback_wait(i) for i in range(C)]      # allocate staging, warm up
t0 = time.perf_counter()
gen.update_all(UPDATE_DELTA, params); gen.readback_begin(range(C))
for _ in range(K - 1):
    gen.update_all(UPDATE_DELTA, params)      # overlaps the copy of the previous tick
    for i in range(C): gen.readback_wait(i)   # previous tick has landed
    gen.readback_begin(range(C))
for i in range(C): gen.readback_wait(i)
dt = time.perf_counter() - t0
out["pipelined_readback"] = {"maps_per_s": round(K * C / dt, 1), "GBps": round(K * C * layer_mb / dt / 1e3, 2), "ms_per_tick": round(dt / K * 1e3, 4)}

# b) one cascade per frame + its readback
t0 = time.perf_counter(); frames = 0
for _ in range(max(1, K // C)):
    gen.update(UPDATE_DELTA, params)
    while gen.pass_num_cascades_remaining:
        idx = gen.pass_num_cascades_remaining - 1
        gen._process(); gen.readback_begin([idx])
        if frames: gen.readback_wait(prev)
        prev = idx; frames += 1
gen.readback_wait(prev)
dt = time.perf_counter() - t0
out["one_cascade_per_frame"] = {"frames_per_s": round(frames / dt, 1), "ms_per_frame": round(dt / frames * 1e3, 4)}

# c) synchronous pageable copy
t0 = time.perf_counter()
for k in range(20): gen.get_maps(k % C)
dt = time.perf_counter() - t0
out["sync_get_maps"] = {"maps_per_s": round(20 / dt, 1), "GBps": round(20 * layer_mb / dt / 1e3, 2)}
print(json.dumps(out))
