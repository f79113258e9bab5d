#!/usr/bin/env python3
"""Prints the parity margins of the HIP path against the oracle (per channel), for a few configurations."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA
for n, ids in [(256, [0, 1, 2, 3]), (512, [2, 4]), (1024, [2]), (1024, [0, 1, 2, 3]), (2048, [1])]:
    gen = WaveGenerator(); gen.map_size = n; gen.debug_f32 = True; gen.init_gpu(max(2, len(ids)))
    params = [WaveCascadeParameters(**cascade_preset(ci)) for ci in ids]
    og = H.oracle_generator(n, ids)
    for frame in range(3):
        gen.update_all(UPDATE_DELTA, params); og.update_all(UPDATE_DELTA); gen.sync()
    for i in range(len(ids)):
        f32, ref = gen.get_maps_f32(i), og.f32(i)
        errs = " ".join(f"{name}:{H.relmax(f32[..., c], ref[..., c]):.1e}" for c, name in enumerate(H.CHANNELS) if name != "foam")
        disp, norm = gen.get_maps(i)
        print(n, ids[i], gen.last_kernel_family(), errs, "fp16 disp %.2f norm %.2f" % (H.fp16_close(disp, og.displacement(i)), H.fp16_close(norm[..., :3], og.normal(i)[..., :3])))

# ow_run's tick groups (small batches): six ticks through k_tick_group_c_lp against the oracle's six
for n, ids in [(256, [0, 1, 2, 3]), (1024, [2])]:
    gen = WaveGenerator(); gen.map_size = n; gen.debug_f32 = True; gen.init_gpu(max(2, len(ids)))
    params = [WaveCascadeParameters(**cascade_preset(ci)) for ci in ids]
    og = H.oracle_generator(n, ids)
    gen.run(UPDATE_DELTA, params, 6); gen.sync()
    for _ in range(6):
        og.update_all(UPDATE_DELTA)
    for i in range(len(ids)):
        f32, ref = gen.get_maps_f32(i), og.f32(i)
        errs = " ".join(f"{name}:{H.relmax(f32[..., c], ref[..., c]):.1e}" for c, name in enumerate(H.CHANNELS) if name != "foam")
        disp, norm = gen.get_maps(i)
        print(n, ids[i], gen.last_kernel_family(), "(6 ticks)", errs, "fp16 disp %.2f norm %.2f" % (H.fp16_close(disp, og.displacement(i)), H.fp16_close(norm[..., :3], og.normal(i)[..., :3])))
# against the bytes the reference's own shaders produced (tests/golden/ref_n1024_c2_f2.npz), the fixture's cascade as one of a batch of four
z = np.load(os.path.join(ROOT, "tests", "golden", "ref_n1024_c2_f2.npz"))
gen = WaveGenerator(); gen.map_size = 1024; gen.init_gpu(4)
params = [WaveCascadeParameters(**cascade_preset(c)) for c in (0, 2, 1, 3)]
for _ in range(int(z["frames"])):
    gen.update_all(float(z["delta"]), params)
gen.sync()
disp, norm = gen.get_maps(1)
st = int(z["row_stride"])
print("1024 cascade 2 vs reference-shader fixture:", gen.last_kernel_family(), "fp16 disp %.2f norm %.2f (of 1 ulp + 1e-5 max)" % (
    H.fp16_close(disp[::st], z["displacement"]), H.fp16_close(norm[::st][..., :3], z["normal"][..., :3])),
    "maps bit-equal on %.4f %% of the fixture's texel channels" % (100.0 * np.mean(disp[::st].view(np.uint16) == z["displacement"])))
