#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REFERENCE'S OWN shaders (oracle/_ref/libglsl_ref.so: the .glsl sources
of /root/reference/assets/shaders/compute compiled as C++ through oracle/glsl_shim.h).  Run where the reference
checkout exists:   make -C oracle ref && python tests/golden/make_golden.py
The fixtures are what tests/test_golden.py holds the oracle (bit-exact) and the HIP path (tolerance) against;
on the GPU box /root/reference does not exist and only these files travel.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from godotoceanwaves_amd.presets import UPDATE_DELTA, cascade_preset  # noqa: E402
from oracle import ref as R  # noqa: E402

CASES = [  # (map_size, cascade preset id, frames, row stride of the stored maps)
    (128, 0, 3, 1),
    (128, 2, 3, 1),
    (256, 1, 2, 4),
    # the reference's default size (water.gd:38) and the one below it: 1024-invocation workgroups of fft_compute.glsl
    # run as fibers, ~40 s per frame at 1024^2 -- generated once, the files are committed
    (512, 3, 2, 16),
    (1024, 2, 2, 64),
    (1024, 0, 1, 64),
    # three frames: what ow_run(3) leaves behind after one ordinary tick and two merged launches (tick groups / tick pairs)
    (512, 3, 3, 16),
    (1024, 2, 3, 64),
    # the other two cascades of the headline batch (1024^2 x 4 = presets 0..3): with ref_n1024_c0_f3 / c1 / c2 / c3 every cascade of the
    # bench configuration is held to bytes of the reference's own shaders after ow_run(3)
    (1024, 1, 3, 64),
    (1024, 3, 3, 64),
    (1024, 0, 3, 64),
]


def main():
    if not R.available():
        raise SystemExit("oracle/_ref/libglsl_ref.so missing: run `make -C oracle ref` where /root/reference exists")
    force = "--force" in sys.argv
    for n, ci, frames, stride in CASES:
        out = os.path.join(HERE, f"ref_n{n}_c{ci}_f{frames}.npz")
        if os.path.exists(out) and not force:   # committed fixtures are only rewritten on request
            print(out, "exists (use --force to regenerate)")
            continue
        rc = R.RefCascade(n, cascade_preset(ci))
        for _ in range(frames):
            rc.update(UPDATE_DELTA)
        np.savez_compressed(
            out, map_size=n, cascade=ci, frames=frames, row_stride=stride, delta=UPDATE_DELTA,
            spectrum_rows=rc.spectrum[::max(stride, 8)].copy(),          # FP32 h0 texels (subset of rows)
            intermediate_rows=rc.intermediate[:, ::max(stride, 8)].copy(),  # fft_buffer half 0 after transpose (last frame)
            displacement=rc.displacement[::stride].copy(),                # RGBA16F bits after the last frame
            normal=rc.normal[::stride].copy())
        print(out, os.path.getsize(out) // 1024, "KiB")


if __name__ == "__main__":
    main()
