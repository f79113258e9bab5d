#!/usr/bin/env python3
"""What a region of K ticks costs per tick with the two chains, on a stream of the CALLER's (torch's, as bench.py does: one ow_run, then a device-wide synchronize)
or on the context's own (ow_sync), against OW_FLAG_SINGLE_STREAM; each variant in its own process:   python scripts/chain_regions.py [map_size = 1024] [cascades = 4]"""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
KS = [5, 10, 20, 40, 200]
def child(n, count, single, own, calls=False):
    import torch
    from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA
    g = WaveGenerator(); g.map_size = n; g.single_stream = single
    if not own:
        st = torch.cuda.Stream(); g.stream = st.cuda_stream
    g.init_gpu(max(2, count))
    p = [WaveCascadeParameters(**cascade_preset(i)) for i in range(count)]
    sync = g.sync if own else torch.cuda.synchronize
    if calls:
        run_ = g.run
        def as_calls(delta, params, k):
            for _ in range(k): g.update_all(delta, params)
        g.run = as_calls
    for _ in range(30): g.run(UPDATE_DELTA, p, 100)
    sync()
    out = []
    for K in KS:
        reps = max(20, 6000 // K)
        for _ in range(5): g.run(UPDATE_DELTA, p, K); sync()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); g.run(UPDATE_DELTA, p, K); sync(); ts.append((time.perf_counter() - t0) / K * 1e6)
        out.append(float(np.median(ts)))
    print(" ".join(f"{v:7.2f}" for v in out), flush=True)
if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] == "1", sys.argv[5] == "1", len(sys.argv) > 6 and sys.argv[6] == "1"); sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    variants = [("one stream, caller's stream", {}, 1, 0, 0), ("two chains, caller's stream", {}, 0, 0, 0), ("two chains, own stream + ow_sync", {}, 0, 1, 0),
                ("one stream, caller's stream, one ow_update_all per tick", {}, 1, 0, 1), ("two chains, caller's stream, one ow_update_all per tick", {}, 0, 0, 1)]
    # (round 6 also ran knob variants -- no join for a caller's stream, the second chain delayed at the fork: profiles/r06_chain_region_cost.txt holds the patch and the figures)
    print(f"{n}^2 x {count}: median us per tick in regions of K = {KS} ticks (one ow_run + synchronisation each)")
    for rnd in range(2):
        for name, env, single, own, calls in variants:
            e = dict(os.environ); e.update(env)
            r = subprocess.run([sys.executable, __file__, "--child", str(n), str(count), str(single), str(own), str(calls)], env=e, capture_output=True, text=True)
            print(f"{name:60s}: {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}", flush=True)
