#!/usr/bin/env python3
"""What an IRREGULAR tick-by-tick caller pays for the look-ahead: ow_update_all (OW_FLAG_RUN_AS_CALLS) and the reference's schedule
(OW_FLAG_RUN_AS_REFERENCE_SCHEDULE) with the delta switching between two values every k ticks (OW_DEBUG_RUN_DELTA_CHANGE_EVERY, read by
ow_create), against the same calls with one launch per pass (OW_FLAG_NO_TICK_GROUPS), whose cost does not depend on the cadence.
    python scripts/lookahead_misses.py [n:c ...]   us per tick, median of 7 x 400 ticks, one process per cell"""
# NOTE (round 5): the OW_DEBUG_* variables are read only by a library built with -DOW_MEASUREMENT_KNOBS:
#   scripts/build_variant.sh knobs -DOW_MEASUREMENT_KNOBS ;  OCEAN_WAVES_LIB=godotoceanwaves_amd/csrc/build/variants/knobs.so python scripts/<this>.py
# (the work-item forms of the tick groups are ow_config flags now: WaveGenerator.group_forms)
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if __name__ == "__main__":
    for cfg in sys.argv[1:] or ["256:4", "512:4", "1024:1", "1024:4"]:
        n, c = cfg.split(":")
        for mode, base in (("calls", "nomerge"), ("reference", "reference_nomerge")):
            cells = []
            for label, m, k in [("one launch per pass", base, "0"), ("regular", mode, "0")] + [(f"delta changes every {k}", mode, k) for k in ("1", "2", "3", "5", "9")]:
                env = dict(os.environ, OW_DEBUG_RUN_DELTA_CHANGE_EVERY=k)
                r = subprocess.run([sys.executable, os.path.join(HERE, "lookahead_ab.py"), "--child", n, c, m], env=env, capture_output=True, text=True)
                cells.append(f"{label}: {(r.stdout.strip() or r.stderr.strip()[-120:])}")
            print(f"{n}^2 x {c} {'ow_update_all' if mode == 'calls' else 'update + process per cascade'}   " + "  |  ".join(cells), flush=True)
