"""A frozen slice of scripts/fuzz_parity.py: random parameter records over the reference's exported ranges, random seeds, batch shapes,
schedules (update_all / run / update + one cascade per frame), deltas and start times -- the HIP path against the oracle, FP32 channels
within 1e-4 (foam within two FP16 steps) and the RGBA16F maps exactly the RTE quantisation of those channels.  The script itself ran ~270
cases in round 3: every FP32 channel inside 1e-4 (worst 3.5e-5, a 23 : 1 tile an hour into a session; typical 5e-6)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
import fuzz_parity as F  # noqa: E402

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(2024)
    cases = [F.draw_case(rng, sizes=(128, 256, 256, 512)) for _ in range(12)]
    # round 4: the records are FP64 and un-rounded (what wave_generator.gd:69-70 sees), and three frozen cases at 2048^2, the only size
    # that reaches the split-plan pass 1 (k_pass1c_split), one per schedule
    big = np.random.default_rng(4096)
    for schedule in ("update_all", "run", "process"):
        case = F.draw_case(big, sizes=(2048,))
        case["schedule"], case["frames"] = schedule, max(2, min(case["frames"], 3))
        cases.append(case)
    return cases


@pytest.mark.parametrize("case", _cases(), ids=lambda c: f"{c['n']}x{len(c['records'])}_{c['schedule']}_{c['frames']}")
def test_random_records_shapes_and_schedules_match_the_oracle(case):
    for r in case["records"]:   # the draw is FP64: at least the wind speed of a record is not an FP32 value (unless it sits at the clamp)
        assert r["wind_speed"] == 1e-4 or float(np.float32(r["wind_speed"])) != r["wind_speed"]
    worst, family, bad = F.run_case(case)
    assert bad is None, (bad, family, case)
    assert worst < 1e-4
