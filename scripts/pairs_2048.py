#!/usr/bin/env python3
"""2048^2: ow_run's tick pairs (k_tick_pair_c_split) tick-major against cascade-major in blocks of 8 / 64 ticks, and one launch per pass.
One process per variant (the knobs are read by ow_create).   python scripts/pairs_2048.py [n:c ...]   us per tick, median (min) of 7 x 200 ticks"""
# NOTE (round 5): the OW_DEBUG_* variables are read only by a library built with -DOW_MEASUREMENT_KNOBS:
#   scripts/build_variant.sh knobs -DOW_MEASUREMENT_KNOBS ;  OCEAN_WAVES_LIB=godotoceanwaves_amd/csrc/build/variants/knobs.so python scripts/<this>.py
# (the work-item forms of the tick groups are ow_config flags now: WaveGenerator.group_forms)
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(n, c, merged):
    import hashlib
    from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA
    gen = WaveGenerator()
    gen.map_size, gen.tick_groups = n, merged
    gen.init_gpu(max(2, c))
    params = [WaveCascadeParameters(**cascade_preset(i)) for i in range(c)]
    gen.run(UPDATE_DELTA, params, 600)
    gen.sync()
    samples = []
    for _ in range(7):
        t0 = time.perf_counter()
        gen.run(UPDATE_DELTA, params, 200)
        gen.sync()
        samples.append((time.perf_counter() - t0) / 200 * 1e6)
    h = hashlib.sha1(b"".join(gen.get_maps(i)[1].tobytes() for i in range(c))).hexdigest()[:10]
    print(f"{statistics.median(samples):8.2f} ({min(samples):7.2f}) {gen.last_kernel_family():18s} {h}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] == "1")
        sys.exit(0)
    cfgs = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(2048, c) for c in (1, 2, 3, 4, 6, 8)]
    for n, c in cfgs:
        row = []
        variants = [("one launch per pass", "0", None, False), ("pairs tick-major", "1", "1", False), ("pairs cascade-major x8", "1", "8", False), ("x64", "1", "64", False)]
        if n <= 1024:  # multi-batch ticks below 2048^2: also full-size batches where the runtime would halve them
            variants += [("x64, full-size batches", "1", "64", True), ("tick-major, full-size batches", "1", "1", True)]
        for label, merged, block, full in variants:
            env = dict(os.environ)
            env.pop("OW_DEBUG_PAIR_TICK_BLOCK", None)
            env.pop("OW_DEBUG_PAIR_FULL", None)
            if full:
                env["OW_DEBUG_PAIR_FULL"] = "1"
            if block:
                env["OW_DEBUG_PAIR_TICK_BLOCK"] = block
            r = subprocess.run([sys.executable, __file__, "--child", str(n), str(c), merged], env=env, capture_output=True, text=True)
            row.append(f"{label}: {r.stdout.strip() or r.stderr.strip()[-300:]}")
        print(f"{n}^2 x {c}\n    " + "\n    ".join(row), flush=True)
