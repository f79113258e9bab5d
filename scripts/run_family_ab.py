#!/usr/bin/env python3
"""ow_run per kernel family: the runtime's own choice (tick groups on the layer-parallel compact family for small ticks) against the compact
family pinned (tick pairs).   python scripts/run_family_ab.py [n:c ...]   us per tick, median of 7 x 400 ticks, one process per cell"""
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(n, c, kernels):
    from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA
    gen = WaveGenerator()
    gen.map_size = n
    gen.kernels = None if kernels == "auto" else kernels
    gen.init_gpu(max(2, c))
    params = [WaveCascadeParameters(**cascade_preset(i)) for i in range(c)]
    gen.run(UPDATE_DELTA, params, 2000)
    gen.sync()
    samples = []
    for _ in range(7):
        t0 = time.perf_counter()
        gen.run(UPDATE_DELTA, params, 400)
        gen.sync()
        samples.append((time.perf_counter() - t0) / 400 * 1e6)
    print(f"{statistics.median(samples):7.2f} ({gen.last_kernel_family()}, {gen.tick_group_depth()} ticks per launch)")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
        sys.exit(0)
    for cfg in sys.argv[1:] or ["256:8", "512:3", "512:4", "512:5", "512:6", "1024:1"]:
        n, c = cfg.split(":")
        cells = []
        for kernels in ("auto", "compact"):
            r = subprocess.run([sys.executable, __file__, "--child", n, c, kernels], capture_output=True, text=True)
            cells.append(f"{kernels}: {r.stdout.strip() or r.stderr.strip()[-160:]}")
        print(f"{n}^2 x {c}   " + "   |   ".join(cells), flush=True)
