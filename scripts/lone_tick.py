#!/usr/bin/env python3
"""A/B of the lone-tick launch shapes (what ow_update_all / ow_process callers get: no look-ahead across ticks), one process per
variant (the knob is read by ow_create):
    OW_DEBUG_SPLIT_TICK=-1  two launches per batch (k_pass1c, k_pass2c)              -- ow_run degraded to a loop of lone ticks
    OW_DEBUG_SPLIT_TICK=11  three launches: [p1 A] [p2 A + p1 B] [p2 B], A = B = half the batch
    OW_DEBUG_SPLIT_TICK=12  the same with A = one cascade
    unset                   ow_run's merged launches across ticks (tick pairs / groups), for reference
  python scripts/lone_tick.py [n cascades]...    prints us per tick, median of 7 regions of 400 ticks"""
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(n, c):
    from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA
    gen = WaveGenerator()
    gen.map_size = n
    gen.init_gpu(max(2, c))
    params = [WaveCascadeParameters(**cascade_preset(i)) for i in range(c)]
    gen.run(UPDATE_DELTA, params, 3000)
    gen.sync()
    samples = []
    for _ in range(7):
        t0 = time.perf_counter()
        gen.run(UPDATE_DELTA, params, 400)
        gen.sync()
        samples.append((time.perf_counter() - t0) / 400 * 1e6)
    d, _ = gen.get_maps(0)
    import hashlib
    print(f"{statistics.median(samples):.2f} {min(samples):.2f} {gen.last_kernel_family()} {hashlib.sha1(d.tobytes()).hexdigest()[:10]}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]))
        sys.exit(0)
    cfgs = [(1024, 4), (1024, 2), (1024, 3), (1024, 6), (1024, 8), (512, 8)]
    if len(sys.argv) > 2:
        cfgs = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
    for n, c in cfgs:
        row = []
        for knob in ("-1", "11", "12", None):
            env = dict(os.environ)
            env.pop("OW_DEBUG_SPLIT_TICK", None)
            if knob:
                env["OW_DEBUG_SPLIT_TICK"] = knob
            r = subprocess.run([sys.executable, __file__, "--child", str(n), str(c)], env=env, capture_output=True, text=True)
            row.append(f"{knob or 'merged'}: {r.stdout.strip() or r.stderr.strip()[-200:]}")
        print(f"{n}^2 x {c}   " + "   |   ".join(row), flush=True)
