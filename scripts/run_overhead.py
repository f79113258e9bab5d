#!/usr/bin/env python3
"""What one ow_run CALL costs beyond its ticks: regions of exactly K ticks between two synchronisations for K = 5 .. 640, per way of driving the
boundary; a straight-line fit region(K) = a + b K gives the fixed cost a per call (launch latency + wake-up of the synchronising host + whatever
the call's own first / last launches waste) and the steady rate b per tick.  (Round 5: the driver times 20-tick regions; ow_run's ordinary first
tick and the two half-filled launches at the ends of a run of tick pairs showed up there as 2 us per tick -- now a seamless stream for single-batch
ticks, ow_runtime.hip ow_run.)
    python scripts/run_overhead.py [n:c ...]"""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA
from bench import Driver

cases = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(1024, 4), (1024, 8), (2048, 4), (256, 4)]
stream = torch.cuda.Stream()
KS = (5, 10, 20, 40, 80, 160, 320, 640)
for n, C in cases:
    def ctx(**attrs):
        g = WaveGenerator(); g.map_size = n; g.stream = stream.cuda_stream
        for k, v in attrs.items():
            setattr(g, k, v)
        g.init_gpu(max(2, C))
        d = Driver(g, [WaveCascadeParameters(**cascade_preset(i)) for i in range(C)])
        d.run(UPDATE_DELTA, 300); torch.cuda.synchronize()
        return d
    drivers = {"ow_run": ctx(), "one ow_update_all per tick (run_as_calls)": ctx(run_as_calls=True), "one launch per pass": ctx(tick_groups=False)}
    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:
        for d in drivers.values():
            d.run(UPDATE_DELTA, 100)
        torch.cuda.synchronize()
    print(f"== {n}^2 x {C}")
    for name, d in drivers.items():
        med = {}
        for K in KS:
            if K * n * n * C > (1 << 33):
                continue
            reps = max(5, min(60, int(0.05 / (K * 60e-6 * max(1, n * n * C / (4 << 20))))))
            s = []
            for _ in range(reps):
                torch.cuda.synchronize(); t0 = time.perf_counter(); d.run(UPDATE_DELTA, K); torch.cuda.synchronize()
                s.append((time.perf_counter() - t0) * 1e6)
            med[K] = statistics.median(s)
        ks = sorted(med)
        mx, my = statistics.mean(ks), statistics.mean(med[k] for k in ks)
        b = sum((k - mx) * (med[k] - my) for k in ks) / sum((k - mx) ** 2 for k in ks)
        a = my - b * mx
        print(f"  {name:45s}: per call {a:7.1f} us + {b:7.3f} us per tick   (" + "  ".join(f"K={k}: {med[k] / k:.2f}" for k in ks) + f" us per tick; {d.gen.last_kernel_family()})", flush=True)
    for d in drivers.values():
        d.free()
