#!/usr/bin/env python3
"""ow_run in tick groups: ticks per launch (OW_DEBUG_TICK_GROUP_DEPTH, read by ow_create) against the runtime's own rule.
    python scripts/group_depth.py [n:c:d,d,.. ...]   us per tick, median of 7 x 400 ticks, one process per cell"""
# NOTE (round 5): the OW_DEBUG_* variables are read only by a library built with -DOW_MEASUREMENT_KNOBS:
#   scripts/build_variant.sh knobs -DOW_MEASUREMENT_KNOBS ;  OCEAN_WAVES_LIB=godotoceanwaves_amd/csrc/build/variants/knobs.so python scripts/<this>.py
# (the work-item forms of the tick groups are ow_config flags now: WaveGenerator.group_forms)
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if __name__ == "__main__":
    cfgs = sys.argv[1:] or ["256:8:4,6,8", "512:2:4,6,8", "512:4:2,4,6,8", "512:5:2,3,4,6", "512:6:2,3,4,6", "1024:1:2,4,6,8"]
    for cfg in cfgs:
        n, c, depths = cfg.split(":")
        cells = []
        for d in ["auto"] + depths.split(","):
            env = dict(os.environ)
            env.pop("OW_DEBUG_TICK_GROUP_DEPTH", None)
            if d != "auto":
                env["OW_DEBUG_TICK_GROUP_DEPTH"] = d
            r = subprocess.run([sys.executable, os.path.join(HERE, "run_family_ab.py"), "--child", n, c, "auto"], env=env, capture_output=True, text=True)
            cells.append(f"{d}: {(r.stdout.strip() or r.stderr.strip()[-160:])}")
        print(f"{n}^2 x {c}   " + "   |   ".join(cells), flush=True)
