#!/usr/bin/env python3
"""Does the tick time depend on WHERE a context's buffers land?  Several contexts of the same configuration are created one after
the other (the earlier ones stay alive, so every new one gets different memory) and each is timed, merged launches and one launch
per pass; then the same with torch-owned maps and a torch stream (what bench.py does)."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA
n, c = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1024:4").split(":"))
K = 2000
def timed(g, p, reps=3):
    best = []
    for _ in range(reps):
        t0 = time.perf_counter(); g.run(UPDATE_DELTA, p, K); g.sync(); best.append((time.perf_counter() - t0) / K * 1e6)
    return statistics.median(best)
keep = []
for style in ("own", "torch"):
    for it in range(5):
        row = []
        for merged in (True, False):
            g = WaveGenerator(); g.map_size = n; g.tick_groups = merged
            if style == "torch":
                s = torch.cuda.Stream()
                d = torch.zeros((max(2, c), n, n, 4), dtype=torch.float16, device="cuda"); m = torch.zeros_like(d)
                torch.cuda.synchronize()
                g.stream = s.cuda_stream; g.external_maps = (d.data_ptr(), m.data_ptr()); keep += [s, d, m]
            g.init_gpu(max(2, c))
            p = [WaveCascadeParameters(**cascade_preset(i)) for i in range(c)]
            g.run(UPDATE_DELTA, p, 600); g.sync()
            row.append(timed(g, p))
            keep.append(g)
        print(f"{style:5s} context pair {it}: merged {row[0]:6.2f} us   one launch per pass {row[1]:6.2f} us", flush=True)
