#!/usr/bin/env python3
"""ow_update_all tick by tick (OW_FLAG_RUN_AS_CALLS) on the layer-parallel compact family: ticks of pass 1 per look-ahead launch
(OW_DEBUG_LOOKAHEAD_DEPTH) x form of the pass-1 items (OW_DEBUG_TICK_GROUP_P1).   python scripts/lookahead_depth.py [n:c ...]   us per tick"""
# NOTE (round 5): the OW_DEBUG_* variables are read only by a library built with -DOW_MEASUREMENT_KNOBS:
#   scripts/build_variant.sh knobs -DOW_MEASUREMENT_KNOBS ;  OCEAN_WAVES_LIB=godotoceanwaves_amd/csrc/build/variants/knobs.so python scripts/<this>.py
# (the work-item forms of the tick groups are ow_config flags now: WaveGenerator.group_forms)
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if __name__ == "__main__":
    cfgs = sys.argv[1:] or ["256:4", "256:8", "512:1", "512:2", "512:3", "512:4", "512:6", "1024:1"]
    for cfg in cfgs:
        n, c = cfg.split(":")
        cells = []
        for form in ("lp", "compact"):
            for depth in (os.environ.get("DEPTHS") or "1 2 3 4").split():
                env = dict(os.environ, OW_DEBUG_TICK_GROUP_P1=form, OW_DEBUG_LOOKAHEAD_DEPTH=depth)
                r = subprocess.run([sys.executable, os.path.join(HERE, "lookahead_ab.py"), "--child", n, c, "calls"], env=env, capture_output=True, text=True)
                cells.append(f"{form} x{depth}: {(r.stdout.strip() or r.stderr.strip()[-120:]).split(' (')[0].strip()}")
        print(f"{n}^2 x {c}   " + "  |  ".join(cells), flush=True)
