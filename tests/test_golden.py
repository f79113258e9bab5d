"""The HIP path against the committed golden vectors (tests/golden/*.npz = outputs of the reference's own
shaders, see tests/golden/make_golden.py).  Tolerances: RGBA16F maps <= 1 fp16 ulp, foam (recurrent FP16 state)
within one FP16 step of [0,1], fft_buffer after pass 1 <= 1e-5 and spectrum <= 2e-5 max-norm relative."""
import glob
import os

import numpy as np
import pytest

import helpers as H
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator
from godotoceanwaves_amd.presets import cascade_preset

pytestmark = pytest.mark.gpu
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_*.npz")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_hip_path_matches_reference_shader_outputs(path):
    z = np.load(path)
    n, ci, frames, stride = int(z["map_size"]), int(z["cascade"]), int(z["frames"]), int(z["row_stride"])
    gen = WaveGenerator()
    gen.map_size = n
    gen.init_gpu(2)
    params = [WaveCascadeParameters(**cascade_preset(ci))]
    for _ in range(frames):
        gen.update_all(float(z["delta"]), params)
    gen.sync()
    sub = max(stride, 8)
    h0, _ = gen.get_spectrum(0)
    assert H.relmax(h0[::sub], z["spectrum_rows"]) < 2e-5
    assert H.relmax(gen.get_intermediate(0)[:, ::sub], z["intermediate_rows"]) < 1e-5
    disp, norm = gen.get_maps(0)
    assert H.fp16_close(disp[::stride], z["displacement"]) <= 1.0
    assert H.fp16_close(norm[::stride][..., :3], z["normal"][..., :3]) <= 1.0
    foam, foam_ref = norm[::stride][..., 3].view(np.float16).astype(np.float64), z["normal"][..., 3].view(np.float16).astype(np.float64)
    assert np.abs(foam - foam_ref).max() <= H.TOL_FOAM_ABS
