"""The FP16 margin of the shipped build, guarded by the suite: `helpers.fp16_close` allows one FP16 ulp PLUS 1e-5 of the
channel maximum -- the floor exists because an ulp shrinks at zero crossings while the FP32 error of a long transform does not.  How many
texel-channels actually need that floor is a property of the kernels, and a change that doubled it would still pass `fp16_close`.  Here the
shares themselves are bounded, per BASELINE configuration, through the call bench.py times (ow_run on a production context):
  * nothing beyond (1 ulp + floor),
  * at most 0.5 % of the FP16 texel-channels of both maps need the floor (measured 0.04 % at 256^2 ... 0.24 % at 2048^2),
  * at least 97 % are bit-equal to the oracle's,
  * and the FP32 channels of a debug context run side by side (bit-identical maps) stay below north_star's 1e-4.
The measurement is scripts/parity_margins.py's (which prints the table kept under profiles/)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def _margins():
    spec = importlib.util.spec_from_file_location("parity_margins", os.path.join(os.path.dirname(__file__), "..", "scripts", "parity_margins.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("n,count", [(256, 4), (1024, 4), (1024, 8), (2048, 4)], ids=["C2_256x4", "C3_1024x4", "C4_total_1024x8", "C5_2048x4"])
def test_fp16_margins_of_the_baseline_configs(n, count):
    pm = _margins()
    assert (n, count) in pm.CONFIGS
    m = pm.measure(n, count)
    print(f"{n}^2 x {count} ({m['family']}): bit-equal {m['bit_equal']*100:.3f} %, one ulp {m['one_ulp']*100:.3f} %, floor {m['floor']*100:.4f} %, "
          f"beyond {m['beyond']*100:.4f} %, worst ratio {m['worst_ratio']:.2f}, FP32 {m['worst_f32']:.1e}")
    assert m["beyond"] == 0.0 and m["worst_ratio"] <= 1.0
    assert m["floor"] <= 0.005
    assert m["bit_equal"] >= 0.97
    assert m["worst_f32"] < 1e-4
