#!/usr/bin/env python3
"""Are two BUILDS of the library bit-identical at the boundary?  Every build (an earlier round's package + library, HEAD, a variant library of HEAD:
the name=dir[:lib] arguments of scripts/ab_rounds.py) runs the same schedule in its own process -- spectrum tick, ow_run, tick-by-tick update_all,
the reference's update + process calls -- and prints one SHA-1 over both RGBA16F maps of every cascade per configuration; the hashes must agree.
    python scripts/hash_maps.py [--configs 256:4,1024:4,...] name=dir[:lib] ..."""
import argparse, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = r'''
import sys, hashlib
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA
for cfg in sys.argv[1].split(","):
    n, c = (int(v) for v in cfg.split(":"))
    g = WaveGenerator(); g.map_size = n; g.init_gpu(max(2, c))
    p = [WaveCascadeParameters(**cascade_preset(i)) for i in range(c)]
    g.run(UPDATE_DELTA, p, 7)
    for _ in range(4):
        g.update_all(UPDATE_DELTA, p)
    g.update(0.03, p)
    for _ in range(c):
        g._process(0.0)
    g.run(UPDATE_DELTA, p, 3)
    g.sync()
    h = hashlib.sha1()
    for i in range(c):
        d, m = g.get_maps(i)
        h.update(d.tobytes()); h.update(m.tobytes())
    print("HASH", cfg, h.hexdigest(), repr(p[-1].time))
    g.free()
'''
ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="256:4,512:2,512:8,1024:1,1024:4,1024:8,2048:1,2048:2,128:2")
ap.add_argument("builds", nargs="+")
a = ap.parse_args()
seen = {}
for b in a.builds:
    name, rest = b.split("=", 1)
    d, _, lib = rest.partition(":")
    env = {**os.environ, "PYTHONPATH": os.path.abspath(os.path.join(ROOT, d))}
    env.pop("OCEAN_WAVES_LIB", None)
    if lib:
        env["OCEAN_WAVES_LIB"] = os.path.abspath(os.path.join(ROOT, lib))
    r = subprocess.run([sys.executable, "-c", DRIVER, a.configs], cwd="/tmp", env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        print(f"{name}: FAILED rc={r.returncode} {r.stderr[-400:]}")
        continue
    for line in r.stdout.splitlines():
        if line.startswith("HASH"):
            _, cfg, hx, t = line.split()
            seen.setdefault(cfg, {})[name] = (hx, t)
bad = 0
for cfg, by in seen.items():
    same = len({v for v in by.values()}) == 1
    bad += not same
    print(f"{cfg:>8}: " + ("IDENTICAL  " if same else "DIFFERENT  ") + "  ".join(f"{k}={v[0][:12]}" for k, v in by.items()))
print("all builds bit-identical on every configuration" if bad == 0 and seen else f"{bad} configuration(s) differ")
sys.exit(1 if bad else 0)
