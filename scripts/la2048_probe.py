#!/usr/bin/env python3
"""would a tick-by-tick caller of ow_update_all at 2048^2 x C gain from a per-cascade look-ahead -- per cascade ONE launch
[pass 2 of (c, t) + speculated pass 1 of (c, t + delta)], tick-major, two scratch slots per cascade?  The runtime's look-ahead is single-batch, so
the launch stream is EMULATED without new runtime code: C contexts of ONE cascade each on one stream (each has the single-batch look-ahead and its
own two-slot scratch), called round-robin once per tick -- exactly the proposed launches in the proposed order -- against one context of C cascades
driven tick by tick (one launch per pass: what such a caller gets today) and against ow_run (cascade-major pairs).
    python scripts/la2048_probe.py [n:c ...]   us per tick of all cascades, median [min..max] of 7 regions, alternating"""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA
from bench import Driver

cases = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(2048, 4), (2048, 2), (1024, 8)]
stream = torch.cuda.Stream()
for n, C in cases:
    def ctx(ids, **attrs):
        g = WaveGenerator(); g.map_size = n; g.stream = stream.cuda_stream
        for k, v in attrs.items():
            setattr(g, k, v)
        g.init_gpu(max(2, len(ids)))
        return Driver(g, [WaveCascadeParameters(**cascade_preset(i)) for i in ids])
    per_batch = 1 if n >= 2048 else 4
    batches = [list(range(b, min(C, b + per_batch))) for b in range(0, C, per_batch)]
    split = [ctx(ids) for ids in batches]            # one context per batch: the emulated per-batch look-ahead
    whole = ctx(list(range(C)))                      # tick by tick, today's path (multi-batch ticks never speculate)
    runner = ctx(list(range(C)))                     # ow_run
    K = 300 if n >= 2048 else 600
    def tick_split():
        for d in split:
            d.update_all(UPDATE_DELTA)
    forms = {"per-batch look-ahead (emulated)": lambda: [tick_split() for _ in range(K)], "update_all, one launch per pass": lambda: [whole.update_all(UPDATE_DELTA) for _ in range(K)],
             "ow_run": lambda: runner.run(UPDATE_DELTA, K)}
    for f in forms.values():
        f(); torch.cuda.synchronize()
    samples = {k: [] for k in forms}
    for rep in range(7):
        for k, f in (list(forms.items()) if rep % 2 == 0 else list(forms.items())[::-1]):
            torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize()
            samples[k].append((time.perf_counter() - t0) / K * 1e6)
    hits = sum(d.hits() for d in split)
    print(f"{n}^2 x {C}: " + "   ".join(f"{k}: {statistics.median(v):.1f} [{min(v):.1f}..{max(v):.1f}] us" for k, v in samples.items()) + f"   (look-ahead hits {hits}, whole-context hits {whole.hits()})", flush=True)
    for d in split + [whole, runner]:
        d.free()
