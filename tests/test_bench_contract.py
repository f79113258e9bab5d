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
    # (the headline configuration also measures roofline.traffic itself -- two rocprofv3 --pmc passes; the other two quote the profiling visit's figure)
    r = run_bench("--steps", "40", "--warmup", "5", "--min-time", "0.05", "--cpu-seconds", "1", *flags, *(() if not flags else ("--no-measure-traffic",)))
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
    assert rf["frac_of_copy_ceiling"] < 1.08   # (the guide's 6.29 TB/s is a DRAM-to-DRAM copy; 226 of the headline's 302 MB per tick are re-read out of the Infinity Cache, which returns ~10 % faster: as two chains the tick runs AT that figure)
    # the interactive path's figure (one launch per pass: what ow_update_all / ow_process callers get) and the residency note
    um, res = rf["unmerged"], rf["residency"]
    assert um["ms_per_step"] >= 0.9 * d["ms_per_step"] and 0.0 < um["frac"] < 0.85 and um["value"] > 0 and len(um["kernels"]) == 2
    assert res["reused_bytes"] == res["spectra_bytes"] + res["intermediate_bytes"] + res["foam_bytes"] and res["infinity_cache_bytes"] == 256 << 20
    # ... and the tick-by-tick caller of ow_update_all (one call per tick, its adaptive look-ahead on): between the two
    uc = rf["update_all_calls"]
    assert uc["lookahead_hit_rate"] > 0.9 and 0.0 < uc["frac"] < 0.85 and uc["ms_per_step"] <= 1.05 * um["ms_per_step"]
    if not flags:  # round 6: the headline's tick-pair launches go out as two chains, and the same regions on ONE stream are timed beside them (same lease): never faster
        assert rf.get("concurrent_launches") == 2 and rf["one_stream"]["ms_per_step"] >= 0.99 * d["ms_per_step"] and 0.0 < rf["one_stream"]["frac"] <= rf["frac"] + 0.01
    else:
        assert "one_stream" not in rf and "concurrent_launches" not in rf
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["value"] > 0
    rs = d["cpu_baseline"]["reference_shaders"]   # the reference's own GLSL on one host core (oracle/_ref, prebuilt): a bounded sample beside the port
    assert "error" in rs or (rs["kind"] == "reference" and rs["cores"] == 1 and 0.0 < rs["value"] < d["cpu_baseline"]["value"])
    if not flags:  # measured by the run itself, and close to the design bytes (72.7 B/texel at the memory side against 72)
        assert rf["traffic_source"].startswith("measured by this run"), rf.get("traffic_measurement_failed")
        # (81 ticks, every launch a full pair: the seamless stream of round 5 -- as two chains, round 6, two launches of half the cascades per tick)
        assert 0.9 < rf["traffic"] / rf["bytes_per_launch"] < 1.15 and rf["traffic_detail"]["full_launch_equivalents"] == 81 * rf.get("concurrent_launches", 1)
    else:
        assert rf["traffic"] is None or "NOT measured by this run" in rf["traffic_source"]
    # round 5: the four ways of driving the boundary are timed interleaved, with the clocks in the record; the scene's own cadence and -- on the
    # headline run -- BASELINE configs C5 / C2 ride in the same line, early in `roofline` (the driver's record keeps its first keys)
    il = d["interleaving"]
    assert il["regions_per_block"] >= 1 and il["blocks_of_value"] >= 3 and len(il["block_medians_ms_per_step"]) >= 3
    assert d["clock_priming"]["probes"] >= 1 and d["clock_priming"]["last_probe_ms_per_step"] > 0
    assert "clocks" in rf and set(rf["clocks"]) >= {"source", "idle_before", "run"}
    if rf["clocks"]["source"]:   # (amdsmi present: the figures are there and plausible)
        assert rf["clocks"]["run"]["sclk_mhz"] is None or 300 <= rf["clocks"]["run"]["sclk_mhz"] <= 3500
    sc = rf["scene_schedule"]
    for key in ("144hz_jitter5pct", "60hz_jitter5pct", "144hz_fixed_clock"):
        e = sc[key]
        assert e["us_per_update"] > 0 and e["maps_per_s"] > 0 and 0.0 <= e["lookahead_hit_rate"] <= 1.0
        assert 0 < e["frame_gpu_us_p50"] <= e["frame_gpu_us_p99"] <= e["frame_gpu_us_max"] and 0.0 < e["frac"] < 0.85
    assert sc["144hz_fixed_clock"]["us_per_update"] <= 1.1 * sc["144hz_jitter5pct"]["us_per_update"]   # (a regular cadence never costs more)
    keys = list(rf)
    assert keys.index("scene_schedule") < 16 and keys.index("clocks") < 16 and keys.index("traffic") < 8
    if not flags:
        # round 6: north_star's whole single-GPU grid rides in the line -- the other eleven configurations of 256^2 .. 2048^2 x {1, 4, 8} as a list at
        # the END of `roofline` (a record that keeps the tail of stdout has it), their fractions as scalars at its front (one that keeps the first scalars)
        oc = {e["workload"]: e for e in rf["other_configs"]}
        assert set(oc) == {f"{gn}^2 x {gc}" for gn in (256, 512, 1024, 2048) for gc in (1, 4, 8)} - {"1024^2 x 4"}
        assert keys.index("other_configs") > keys.index("residency") and keys.index("grid_frac_x1_x4_x8") < 10 and keys.index("c5_2048x4_frac") < 12
        assert oc["2048^2 x 4"]["kernel"] == "k_tick_pair_c_split" and oc["256^2 x 4"]["kernel"] == "k_tick_group_c_lp"
        assert oc["1024^2 x 8"]["kernel"] == "k_tick_pair_c" and oc["1024^2 x 1"]["kernel"] == "k_tick_group_c_lp"
        for e in oc.values():
            assert e["value"] > 0 and 0.0 < e["frac"] < 0.85 and e["repeats"] >= 3 and e["ms_per_step"] > 0
        assert rf["c5_2048x4_frac"] == oc["2048^2 x 4"]["frac"] and rf["c2_256x4_frac"] == oc["256^2 x 4"]["frac"]
        assert len(rf["grid_frac_x1_x4_x8"]) <= 120 and rf["grid_frac_x1_x4_x8"].count("/") == 8 and "-" not in rf["grid_frac_x1_x4_x8"]
    else:
        assert "other_configs" not in rf
    # round 4: the CPU leg runs first, the GPU work is one contiguous stretch and says how long it was
    assert d["timed_region_s"] == d["timed_seconds"] > 0 and d["gpu_phase_s"] > d["timed_region_s"]
    assert d["timed_ticks_per_region"] == 40 and d["regions_per_sync"] == 1 and d["scaling"] == "weak"


@pytest.mark.gpu
def test_two_rank_rehearsal_carries_the_scaling_references(tmp_path):
    """The N > 1 rank code on the ONE GPU of the box (gloo, both ranks on GPU 0: control flow only, the numbers mean nothing): the default
    is the strong-scaling series of BASELINE config C4 (8 cascades shared out: 4 per GPU at N = 2), the gather keeps its cadence inside a
    region of R x K ticks, and the line carries the references that answer the scaling question on its own."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--min-time", "0.05", "--backend", "gloo", "--share-gpu",
                        "--prime-ms", "50"], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["scaling"] == "strong" and d["config"]["cascades_per_gpu"] == 4
    assert "rehearsal" in d["config"] and "cpu_baseline" not in d
    k = d["gather"]["every_ticks"]
    assert k >= 1 and d["timed_ticks_per_region"] >= 4 * k and d["timed_ticks_per_region"] % 20 == 0
    assert d["regions_per_sync"] == d["timed_ticks_per_region"] // 20 == d["gather"]["regions_per_sync"]
    assert abs(d["value"] - 8 * d["timed_ticks_per_region"] / (d["ms_per_step"] * 1e-3 * d["timed_ticks_per_region"])) / d["value"] < 1e-3
    for key in ("no_gather", "gather_every_tick", "per_gpu_alone", "one_gpu_whole_job", "speedup_with_gather", "speedup_no_gather",
                "speedup_vs_one_gpu_with_gather", "speedup_vs_one_gpu_no_gather", "parallel_efficiency", "headline_basis"):
        assert key in d, key
    assert d["one_gpu_whole_job"]["cascades"] == 8 and d["one_gpu_whole_job"]["value"] > 0
    assert d["per_gpu_alone"]["min"] <= d["per_gpu_alone"]["value"] <= d["per_gpu_alone"]["max"]
    assert abs(d["speedup_no_gather"] - d["no_gather"]["value"] / d["per_gpu_alone"]["value"]) < 0.02 * d["speedup_no_gather"]


def test_scene_schedule_of_the_bench_is_the_water_nodes_policy():
    """roofline.scene_schedule drives the boundary call by call (bench.scene_frames); the calls must be exactly what the mirrored ocean node
    (godotoceanwaves_amd.water.Water = water.gd:75-82 + the generator node's _process) issues on the same frame clock: the same update deltas,
    bit for bit, in the same frames, one process call per frame."""
    sys.path.insert(0, ROOT)
    import bench
    from godotoceanwaves_amd.water import Water

    class Calls:
        def __init__(self):
            self.log = []

        def update(self, delta, parameters=None):
            self.log.append(("update", delta))

        def process(self):
            self.log.append(("process",))

        def _process(self, delta=0.0):   # the generator node's per-frame call
            self.log.append(("process",))

        map_size = 0

        def init_gpu(self, n):
            pass

    for hz, jitter, seed in ((144, 0.05, 777), (60, 0.05, 4242), (144, 0.0, 1)):
        frames = 400
        a = Calls()
        updates = bench.scene_frames(a, hz, frames, jitter, seed=seed)
        b = Calls()
        w = Water(generator_factory=lambda: b)
        w.wave_generator = b
        w._parameters = [object()]
        state = seed
        for _ in range(frames):
            state = (state * 1103515245 + 12345) & 0x7FFFFFFF
            w._process((1.0 / hz) * (1.0 + jitter * (state / 0x3FFFFFFF - 1.0)))
        assert a.log == b.log
        got = [e[1] for e in a.log if e[0] == "update"]
        # (the limiter re-arms relative to the frame that fires, water.gd:80, so an update comes every ceil(frame rate / 50) frames:
        #  every third frame at 144 Hz, every second at 60 Hz -- 48 and 30 updates per second, not 50)
        per = -(-hz // 50)
        assert updates == len(got) and frames / (per + 0.5) <= updates <= frames / per + 1
        assert sum(1 for e in a.log if e[0] == "process") == frames
        if jitter:
            assert len(set(got[3:])) > 0.9 * len(got[3:])                                # every update a different delta
        assert all(0.02 - 1e-12 <= d < 0.02 + 1.06 / hz for d in got)


def test_sensor_summary_takes_medians_and_survives_missing_fields():
    sys.path.insert(0, ROOT)
    import bench
    s = bench.Sensors.summary([{"sclk_mhz": 2100.0, "power_w": 1300.0, "temp_mem_c": None}, None, {"sclk_mhz": 2200.0, "power_w": 1340.0, "temp_mem_c": None},
                               {"sclk_mhz": 2150.0, "power_w": None, "temp_mem_c": None}])
    assert s == {"samples": 3, "sclk_mhz": 2150.0, "power_w": 1320.0, "temp_mem_c": None}
    assert bench.Sensors.summary([None, None]) is None
    assert bench.Sensors._num({"a": "N/A", "b": [2100, 2200, "N/A", 65535], "c": 7}, "a", "b", "c") == 2150.0
