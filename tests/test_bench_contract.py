"""bench.py's output contract (the driver parses it): ONE JSON line with the keys the task statement names, `roofline` on the bytes the
launched kernels move, `cpu_baseline` on request; and a loud failure without a GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline"]
ROOFLINE = ["bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "bytes_per_texel", "frac_of_copy_ceiling", "frac_contract_104", "tick"]


def run_bench(*flags):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, cwd=ROOT)


def test_bench_needs_a_gpu_and_says_so():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = run_bench("--steps", "5", "--warmup", "1")
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)


@pytest.mark.gpu
@pytest.mark.parametrize("flags,kernel", [((), "k_tick_pair_c"), (("--map-size", "256"), "k_tick_group_c_lp"), (("--map-size", "2048", "--cascades", "1"), "k_tick_pair_c_split")])
def test_one_json_line_with_the_contract_keys(flags, kernel):
    r = run_bench("--steps", "40", "--warmup", "5", "--min-time", "0.05", "--cpu-seconds", "1", *flags)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in REQUIRED:
        assert k in d, k
    for k in ROOFLINE:
        assert k in d["roofline"], k
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 5 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "maps/s" and d["value"] > 0 and d["repeats"] >= 1 and "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["kernel"].startswith(kernel)
    assert 0.0 < rf["frac"] < 0.85 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3      # a bandwidth, below the copy ceiling
    assert rf["frac_of_copy_ceiling"] < 1.0
    # the interactive path's figure (one launch per pass: what ow_update_all / ow_process callers get) and the residency note
    um, res = rf["unmerged"], rf["residency"]
    assert um["ms_per_step"] >= 0.9 * d["ms_per_step"] and 0.0 < um["frac"] < 0.85 and um["value"] > 0 and len(um["kernels"]) == 2
    assert res["reused_bytes"] == res["spectra_bytes"] + res["intermediate_bytes"] + res["foam_bytes"] and res["infinity_cache_bytes"] == 256 << 20
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["value"] > 0
