"""SURVEY.md 8f N1 as COMPILED, GPU-tested code: examples/water.hpp (the ocean node's update policy, assets/water/water.gd:22-35,
51-54,75-82,84-114) on top of examples/wave_generator.hpp and the real kernels, driven by examples/water_host.cpp through an
irregular frame script -- frame times that jitter, `updates_per_second` changed mid-run (incl. 0 = update every frame, which leaves
leftovers for the next update's flush, wave_generator.gd:94-98), live parameter edits between an update and the frame that
processes the cascade (wind_speed, tile_length, foam_amount: dirty flag -> spectrum regenerated, wave_generator.gd:68-72), a
map_size change (generator rebuilt, water.gd:38-41).  Held against the CPU ORACLE advanced on the same schedule by the Python
mirror of the policy (godotoceanwaves_amd/water.py): the issued update deltas and node times must agree to the last FP64 bit, the
maps within the FP16 tolerance of the parity tests."""
import os
import subprocess

import numpy as np
import pytest

import helpers as H
from godotoceanwaves_amd import WaveCascadeParameters
from godotoceanwaves_amd.presets import DEPTH, cascade_preset
from godotoceanwaves_amd.water import Water
from oracle import oracle as O
from test_c_consumer import PKG, ROOT

SCRIPT = """
params 3
mapsize 256
frame 0.016
frame 0.016
frame 0.0171
frame 0.0302
frame 0.0049
wind 1 9
frame 0.0166
frame 0.0166
frame 0.0166
rate 25
frame 0.011
frame 0.052
frame 0.0166
tile 2 16 24
frame 0.0166
frame 0.009
rate 0
frame 0.02
frame 0.013
foam 0 2.5
frame 0.027
frame 0.0166
rate 60
frame 0.0166
frame 0.0166
frame 0.021
frame 0.0166
frame 0.0166
frame 0.004
frame 0.0166
dump {out}
"""


def build(tmp_path):
    exe = str(tmp_path / "water_host")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "examples"),
                    os.path.join(ROOT, "examples", "water_host.cpp"), "-o", exe, "-L", PKG, "-locean_waves",
                    f"-Wl,-rpath,{PKG}", "-Wl,-rpath-link,/opt/rocm/lib"], check=True)
    return exe


def test_builds_and_fails_loudly_without_a_device(tmp_path):
    import torch
    exe = build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu test")
    script = tmp_path / "s.txt"
    script.write_text("params 2\nframe 0.016\n")
    r = subprocess.run([exe, str(script)], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr and "no CPU fallback" in r.stderr


class OracleWaveGenerator:
    """WaveGenerator's surface (wave_generator.gd:8,17,56-63,90-109) over the CPU oracle: the reference's own schedule"""

    def __init__(self):
        self.map_size, self.g, self.pass_parameters, self.remaining = 0, None, [], 0

    def init_gpu(self, layers):
        self.g = O.Generator(self.map_size, layers, DEPTH)

    def _update(self, i):  # :65-85, with the LIVE parameter object
        p = self.pass_parameters[i]
        p._pack(self.g.params[i])
        self.g.update_cascade(i)
        p.should_generate_spectrum = False  # :72

    def update(self, delta, parameters):
        for i in range(self.remaining):  # :94-98
            self._update(i)
        for p in parameters:  # :101-106
            p.time += delta
            p.foam_grow_rate = delta * p.foam_amount * 7.5
            p.foam_decay_rate = delta * max(0.5, 10.0 - p.foam_amount) * 1.15
        self.pass_parameters, self.remaining = list(parameters), len(parameters)  # :108-109

    def _process(self, delta):  # :56-63
        if self.remaining:
            self._update(self.remaining - 1)
            self.remaining -= 1

    def free(self):
        if self.g:
            self.g.close()
            self.g = None


def run_mirror(script):
    """the same script through the Python mirror of the policy over the oracle; returns (issued lines, water)"""
    w, lines = Water(OracleWaveGenerator), []
    for line in script.strip().splitlines():
        cmd, *a = line.split()
        if cmd == "params":
            c = int(a[0])
            w.set_parameters([WaveCascadeParameters(**{k: v for k, v in cascade_preset(i).items() if k not in ("spectrum_seed", "time")}) for i in range(c)],
                             seeds=[(1000 + 17 * i, -2000 + 31 * i) for i in range(c)])
        elif cmd == "mapsize":
            w.map_size = int(a[0])
        elif cmd == "frame":
            d = w._process(float(a[0]))
            if d is not None:
                lines.append((d, w.time))
        elif cmd == "rate":
            w.updates_per_second = float(a[0])
        elif cmd == "wind":
            w.parameters[int(a[0])].wind_speed = float(a[1])
        elif cmd == "foam":
            w.parameters[int(a[0])].foam_amount = float(a[1])
        elif cmd == "tile":
            w.parameters[int(a[0])].tile_length = (float(a[1]), float(a[2]))
    return lines, w


@pytest.mark.gpu
def test_compiled_water_node_on_the_real_kernels_follows_the_oracle_on_the_same_schedule(tmp_path):
    out = tmp_path / "maps.bin"
    script = SCRIPT.format(out=out)
    sfile = tmp_path / "script.txt"
    sfile.write_text(script)
    r = subprocess.run([build(tmp_path), str(sfile)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got_updates = [tuple(float(v) for v in l.split()[1::2]) for l in r.stdout.splitlines() if l.startswith("update")]
    lines, w = run_mirror(script)
    # the scheduler: which frames issue an update, with which catch-up delta, at which node time -- to the last bit
    assert got_updates == lines and len(lines) >= 12
    tail = {l.split()[0]: l.split()[1:] for l in r.stdout.splitlines() if not l.startswith(("update", "cascade_time", "map_scale"))}
    assert int(tail["remaining"][0]) == w.wave_generator.remaining and int(tail["remaining"][2]) == 2   # map_size change rebuilt the generator
    assert float(tail["remaining"][4]) == w.next_update_time
    times = [(float(l.split()[1]), int(l.split()[3])) for l in r.stdout.splitlines() if l.startswith("cascade_time")]
    assert times == [(p.time, int(p.should_generate_spectrum)) for p in w.parameters]
    scales = [tuple(float(v) for v in l.split()[1:]) for l in r.stdout.splitlines() if l.startswith("map_scale")]
    assert np.allclose(scales, w.map_scales(), rtol=1e-7) and scales[2][1] == pytest.approx(1 / 24)
    # the maps the compiled node + the HIP kernels left behind against the oracle's
    n, c = 256, 3
    raw = np.fromfile(out, np.uint16).reshape(c, 2, n, n, 4)
    og = w.wave_generator.g
    for i in range(c):
        assert H.fp16_close(raw[i, 0], og.displacement(i)) <= 1.0, i
        assert H.fp16_close(raw[i, 1][..., :3], og.normal(i)[..., :3]) <= 1.0, i
        foam, foam_ref = raw[i, 1][..., 3].view(np.float16).astype(np.float64), og.normal(i)[..., 3].view(np.float16).astype(np.float64)
        assert np.abs(foam - foam_ref).max() <= 2 * H.TOL_FOAM_ABS, i
        assert (np.abs(foam - foam_ref) > H.TOL_FOAM_ABS).mean() < 1e-3
    w.wave_generator.free()
