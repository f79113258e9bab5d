#!/usr/bin/env python3
"""ow_update_all tick by tick (OW_FLAG_RUN_AS_CALLS) with its adaptive look-ahead against one launch per pass (OW_FLAG_NO_TICK_GROUPS), and -- for
the layer-parallel compact family, whose look-ahead launch is the group kernel with one tick per side -- how many ticks of pass 1 a launch computes ahead
(OW_DEBUG_LOOKAHEAD_DEPTH) and the two forms of its pass-1 items (OW_DEBUG_TICK_GROUP_P1), both read by ow_create.  One process per variant.   python scripts/lookahead_ab.py [n:c ...]   us per tick, median of 7 x 400"""
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


def child(n, c, mode):
    from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA
    gen = WaveGenerator()
    gen.map_size = n
    gen.tick_groups = mode != "nomerge"
    gen.run_as_calls = mode == "calls"
    gen.run_as_reference = mode in ("reference", "reference_nomerge")
    if mode == "reference_nomerge":
        gen.tick_groups = False
    gen.group_forms = (os.environ.get("OW_DEBUG_TICK_GROUP_P1") or None, None)  # (the script's own convention: the library does not read it)
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
    print(f"{statistics.median(samples):7.2f} (hits {gen.lookahead_stats()[0]}, launches with work ahead {gen.lookahead_stats()[1]})")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
        sys.exit(0)
    cfgs = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(256, 1), (256, 2), (256, 4), (256, 8), (512, 1), (512, 2), (512, 4), (512, 6), (1024, 1), (1024, 2), (1024, 4), (512, 8)]
    for n, c in cfgs:
        row = []
        for label, mode, p1 in (("one launch per pass", "nomerge", None), ("update_all + look-ahead", "calls", None), ("  one tick ahead", "calls", "depth1"), ("  two", "calls", "depth2"),
                                ("  pass-1 items compact", "calls", "compact"), ("ow_run", "run", None),
                                ("update + one process per cascade, one launch per pass", "reference_nomerge", None), ("  with the prefetch of the next cascade", "reference", None)):
            env = dict(os.environ)
            env.pop("OW_DEBUG_TICK_GROUP_P1", None)
            env.pop("OW_DEBUG_LOOKAHEAD_DEPTH", None)
            if p1 and p1.startswith("depth"):
                env["OW_DEBUG_LOOKAHEAD_DEPTH"] = p1[5:]
            elif p1:
                env["OW_DEBUG_TICK_GROUP_P1"] = p1
            r = subprocess.run([sys.executable, __file__, "--child", str(n), str(c), mode], env=env, capture_output=True, text=True)
            row.append(f"{label}: {r.stdout.strip() or r.stderr.strip()[-200:]}")
        print(f"{n}^2 x {c}   " + "   |   ".join(row), flush=True)
