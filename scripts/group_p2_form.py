#!/usr/bin/env python3
"""tick groups: plain pass-2 blocks against the pipelined form (WaveGenerator.group_forms = OW_FLAG_GROUP_P2_PLAIN | _PIPE).  us per tick of ow_run, and whether
the maps after 37 ticks are bit-identical"""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA
cases = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(256, 1), (256, 2), (256, 4), (256, 8), (512, 1), (512, 2), (512, 4), (1024, 1)]
for n, c in cases:
    row, sums = [], []
    for form in ("plain", "pipe"):
        best = 1e9
        gen = WaveGenerator(); gen.map_size = n; gen.group_forms = (None, form); gen.init_gpu(max(2, c))
        params = [WaveCascadeParameters(**cascade_preset(i)) for i in range(c)]
        gen.run(UPDATE_DELTA, params, 37); gen.sync()
        h = hashlib.sha1()
        for i in range(c):
            d, m = gen.get_maps(i)
            h.update(d.tobytes() + m.tobytes())
        sums.append(h.hexdigest()[:12])
        gen.run(UPDATE_DELTA, params, 1500); gen.sync()
        for rep in range(5):
            t0 = time.perf_counter(); gen.run(UPDATE_DELTA, params, 2000); gen.sync(); best = min(best, time.perf_counter() - t0)
        row.append(f"{form} {best/2000*1e6:6.2f} us ({gen.last_kernel_family()[:12]}, depth {gen.tick_group_depth()})")
        gen.free()
    print(f"{n}^2 x {c}  " + "   ".join(row) + ("   identical" if len(set(sums)) == 1 else f"   DIFFERENT {sums}"), flush=True)
