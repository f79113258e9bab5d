#!/usr/bin/env python3
"""Same-lease A/B of DIFFERENT BUILDS of the library -- earlier rounds' closing states (their own package + library, extracted by
`git archive <commit> godotoceanwaves_amd include` into tools/ab_rounds/<name>/ and built there) against HEAD and against variant libraries
of HEAD (scripts/build_variant.sh) -- on the headline configuration: every build runs in its own short process (its ABI and Python mirror
differ from HEAD's), the processes ALTERNATE (a b c a b c ...), each primes the clocks and times `reps` regions of K ow_run ticks.
   scripts/ab_rounds.py [--cycles 3] [--config 1024:4] name=dir[:lib] ...
     dir = directory that holds the package `godotoceanwaves_amd` (`.` = this tree); lib = optional OCEAN_WAVES_LIB override
   -> per build: us per tick of ow_run, median [min..max] over its regions, per cycle and over all cycles"""
import argparse, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = r'''
import sys, time, statistics
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA
n, c, K, reps, merged = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5] == "1"
g = WaveGenerator(); g.map_size = n
if not merged: g.tick_groups = False
g.init_gpu(max(2, c))
p = [WaveCascadeParameters(**cascade_preset(i)) for i in range(c)]
g.run(UPDATE_DELTA, p, 300); g.sync()
t_end = time.perf_counter() + 1.0
while time.perf_counter() < t_end:
    g.run(UPDATE_DELTA, p, 200); g.sync()
s = []
for r in range(reps):
    t0 = time.perf_counter(); g.run(UPDATE_DELTA, p, K); g.sync(); s.append((time.perf_counter() - t0) / K * 1e6)
print("RESULT", " ".join(f"{v:.3f}" for v in s), g.last_kernel_family())
'''
ap = argparse.ArgumentParser()
ap.add_argument("--cycles", type=int, default=3)
ap.add_argument("--config", default="1024:4")
ap.add_argument("--ticks", type=int, default=2000)
ap.add_argument("--reps", type=int, default=9)
ap.add_argument("--unmerged", action="store_true", help="one launch per pass (OW_FLAG_NO_TICK_GROUPS) instead of ow_run's merged launches")
ap.add_argument("builds", nargs="+")
a = ap.parse_args()
n, c = (int(v) for v in a.config.split(":"))
builds = []
for b in a.builds:
    name, rest = b.split("=", 1)
    d, _, lib = rest.partition(":")
    builds.append((name, os.path.abspath(os.path.join(ROOT, d)), os.path.abspath(os.path.join(ROOT, lib)) if lib else None))
allv = {name: [] for name, _, _ in builds}
for cyc in range(a.cycles):
    order = builds if cyc % 2 == 0 else builds[::-1]
    for name, d, lib in order:
        env = {**os.environ, "PYTHONPATH": d}
        env.pop("OCEAN_WAVES_LIB", None)
        if lib:
            env["OCEAN_WAVES_LIB"] = lib
        r = subprocess.run([sys.executable, "-c", DRIVER, str(n), str(c), str(a.ticks), str(a.reps), "0" if a.unmerged else "1"], cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        if r.returncode != 0 or not line:
            print(f"cycle {cyc} {name}: FAILED rc={r.returncode} {r.stderr[-300:]}", flush=True)
            continue
        tok = line[0].split()
        v = [float(x) for x in tok[1:-1]]
        allv[name] += v
        print(f"cycle {cyc} {name:>14}: {statistics.median(v):7.2f} [{min(v):.2f}..{max(v):.2f}] us per tick  ({tok[-1]})", flush=True)
print(f"== {n}^2 x {c}, {'one launch per pass' if a.unmerged else 'ow_run'}, {a.cycles} cycles x {a.reps} regions x {a.ticks} ticks, alternating processes on one box")
for name, v in allv.items():
    if v:
        print(f"{name:>14}: median {statistics.median(v):7.2f}  min {min(v):7.2f}  max {max(v):7.2f} us per tick")
