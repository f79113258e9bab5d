#!/usr/bin/env python3
"""What the host's wait policy costs a 20-tick region (bench.py's driver-sized regions are ONE ow_run of 20 ticks between two torch.cuda.synchronize()):
the same regions with the HIP runtime's default wait (hipDeviceScheduleAuto) and with hipSetDeviceFlags(hipDeviceScheduleSpin) set before the first
HIP call, each in its own process, alternating.   python scripts/sync_gap_probe.py [ticks]"""
import os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import ctypes, os, sys, time, statistics
mode, K = sys.argv[1], int(sys.argv[2])
if mode != "auto":
    hip = ctypes.CDLL("libamdhip64.so")
    rc = hip.hipSetDeviceFlags(ctypes.c_uint({"spin": 1, "yield": 2, "blocking": 4}[mode]))
    print("hipSetDeviceFlags rc", rc)
sys.path.insert(0, sys.argv[3])
import torch
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA
from bench import Driver
stream = torch.cuda.Stream()
g = WaveGenerator(); g.map_size = 1024; g.stream = stream.cuda_stream; g.init_gpu(4)
d = Driver(g, [WaveCascadeParameters(**cascade_preset(i)) for i in range(4)])
d.run(UPDATE_DELTA, 300); torch.cuda.synchronize()
t_end = time.perf_counter() + 0.8
while time.perf_counter() < t_end:
    d.run(UPDATE_DELTA, 200); torch.cuda.synchronize()
s = []
for _ in range(600):
    torch.cuda.synchronize(); t0 = time.perf_counter(); d.run(UPDATE_DELTA, K); torch.cuda.synchronize(); s.append((time.perf_counter() - t0) / K * 1e6)
long = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); d.run(UPDATE_DELTA, 2000); torch.cuda.synchronize(); long.append((time.perf_counter() - t0) / 2000 * 1e6)
print("RESULT", mode, f"{statistics.median(s):.3f} {min(s):.3f} {statistics.median(long):.3f}")
'''
K = sys.argv[1] if len(sys.argv) > 1 else "20"
for cyc in range(2):
    for mode in ("auto", "spin", "yield") if cyc == 0 else ("spin", "auto"):
        r = subprocess.run([sys.executable, "-c", CHILD, mode, K, ROOT], capture_output=True, text=True, cwd="/tmp", timeout=300)
        out = [l for l in r.stdout.splitlines() if l.startswith(("RESULT", "hipSet"))]
        print(f"cycle {cyc}: " + " | ".join(out) + ("" if r.returncode == 0 else f" rc={r.returncode} {r.stderr[-200:]}"), flush=True)
print("(RESULT mode: median and min us per tick of 600 regions of K ticks; median us per tick of 2000-tick regions)")
