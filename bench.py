#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its headline config, one process per GPU.

  metric : displacement+normal maps/sec, 1024^2 x 4 cascades (1 "map" = one cascade update = one
           displacement layer + one normal/foam layer); achieved HBM GB/s vs peak in `roofline`
  step   : one simulation tick = time-modulate + 2-D IFFT + unpack/foam of all 4 cascades of this rank
           (steady state: the spectra h0 / omega are already resident in HBM; spectrum generation runs once
           during warm-up, like the reference's should_generate_spectrum path, and is reported separately)
  timing : W untimed warm-up ticks, then the region of EXACTLY K ticks bracketed by barrier + synchronize on both
           sides, max over ranks.  The region is repeated (`repeats`) until at least --min-time seconds have been
           timed; `ms_per_step` / `value` are the MEDIAN repeat (a 20-tick region is 1 ms: one sample of it is noise)
  N > 1  : cascades/tiles are independent units: every rank owns its own cascades, no data-path collective.
           The default is BASELINE config C4 as a STRONG-scaling series: --total-cascades 8 shared out evenly, 4 / 2 / 1 1024^2
           cascades per GPU at N = 2 / 4 / 8 ("scaling": "strong"; --cascades C pins C per GPU instead: weak), finished maps
           gathered to rank 0 (the consumer GPU) over RCCL by sharding.MapGatherer -- owned layers only, from a snapshot,
           on a side stream -- INSIDE the timed region, every k ticks, k = the smallest cadence whose gather hides under
           k ticks of compute (measured during warm-up, agreed over ranks; --gather-every k pins it, 0 = once after the
           region).  `value` is that region.  So that one line answers the scaling question on its own, it also carries
             no_gather           the same ticks with no exchange at all (the pure cascade-parallel rate), gather_every_tick (k = 1);
             per_gpu_alone       every rank's rate on its per-GPU config, timed locally, no barrier, no gather;
             one_gpu_whole_job   rank 0 running ALL the job's cascades alone while the other ranks wait (the N = 1 point of the
                                 strong-scaling series, measured in THIS run on THIS node) and
             speedup_*           value and no_gather against both references (speedup_vs_one_gpu_* is north_star's ">= 6x at 8 GPUs").
           A timed region is K = --steps ticks between two barrier + synchronize pairs; where K is shorter than four gather cadences
           the region is R regions of K ticks back to back under ONE outer pair of synchronisations (regions_per_sync), so that the
           gather keeps the cadence the links sustain instead of collapsing to one gather per region.
  N = 1  : four ways of driving the boundary are timed INTERLEAVED, in blocks of ~40 ms that alternate (run, calls, run, unmerged, run,
           reference, ...) after the clocks have been primed until two consecutive 200-tick probes agree within 1 %: `value` (ow_run's merged
           launches, the median of its regions over all its blocks), `roofline.update_all_calls` (one ow_update_all per tick, adaptive
           look-ahead), `roofline.unmerged` (one launch per pass, OW_FLAG_NO_TICK_GROUPS: callers with an irregular cadence),
           `roofline.reference_schedule` (ow_update + one ow_process per cascade, regular cadence) -- so that no figure owes anything to the
           moment it was taken at; `roofline.clocks` carries sclk / mclk / socket power / temperature sampled after every block (amdsmi).
           Every region is EXACTLY K ticks between two synchronisations, enqueued by ONE call of the C-ABI (ow_run through ctypes with a
           pre-packed record array: the Python mirror's per-call packing is not part of the path a C# / C host takes).
           `roofline.scene_schedule`: the reference scene's real cadence -- water.gd:75-82's rate limiter (50 updates/s) over 60 / 144 Hz
           frames, one ow_process per frame, leftovers flushed by the next ow_update, every update a different delta when the frame clock
           jitters -- as us per update, maps/s, look-ahead hit rate and p50 / p99 of the GPU time of one frame's calls.
           `roofline.other_configs`: the other ELEVEN configurations of north_star's single-GPU grid, 256^2 .. 2048^2 x {1, 4, 8} cascades, timed in
           the same process after the headline regions (ow_run regions of exactly `steps_per_region` ticks, same method; one context at a time);
           BASELINE configs C5 (2048^2 x 4, the DRAM-bound one) and C2 (256^2 x 4, the 1000-frame loop) with the longer budget.  The list sits at
           the end of `roofline`; its fractions are repeated as scalars at the front (`grid_frac_x1_x4_x8`, `c5_2048x4_frac`, `c2_256x4_frac`).
           `roofline.residency` says what of the working set fits the 256 MiB Infinity Cache (so a reader knows when "HBM GB/s" is partly
           cache traffic).
  sweep  : --sweep appends one line per BASELINE configuration (256^2 x 4, 1024^2 x {1,4,8}, 2048^2 x 4), each with its
           own roofline and CPU baseline, to --sweep-out (profiles/) and prints the headline line last; --sweep-grid does the same
           over north_star's whole grid, 256^2 .. 2048^2 x {1, 4, 8} cascades.

Launch:  python bench.py [--gpus 1] [--steps K] [--warmup W]
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
                bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import math
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md:35
SINGLE_STREAM = False  # --single-stream (A/B)
COPY_CEILING_GBPS = 6290.0  # the same guide's measured copy rate (profiles/r01_membench_copy_ceiling.txt reproduces it)
# SURVEY.md 8d's contract: algorithmic bytes per texel per cascade-update of a two-pass transform with a four-layer FP32
# intermediate -- pass 1 reads h0 (16) and writes the intermediate (32); pass 2 reads it (32), reads the previous normal
# texel for foam (8) and writes the two RGBA16F maps (8 + 8).
CONTRACT_BYTES = (48, 56)
# Bytes per texel the LAUNCHED kernel family must move, (pass 1, pass 2):
#   four-layer intermediate:  h0(k) 8 + omega 4 read, T 32 written | T 32 + foam 2 read, maps 16 + foam 2 written
#   compact intermediate   :  h0(k) 8 + omega 4 read, T 20 written | T 20 + foam 2 read, maps 16 + foam 2 written
# (the redundant half of the reference's spectrum texel is never stored, foam is a private FP16 plane; PMC counters put
#  the memory-side traffic within 1-2 % of these: profiles/pmc_traffic.json)
FAMILY_BYTES = {"standard": (44, 52), "layer_parallel": (44, 52), "compact": (32, 40), "layer_parallel_compact": (32, 40)}
SUFFIX = {"standard": "", "layer_parallel": "_lp", "compact": "c", "layer_parallel_compact": "c_lp"}
MERGED_KERNEL = {"tick_groups_compact": "k_tick_group_c_lp", "tick_pairs_compact": "k_tick_pair_c"}  # ow_run's launches merged across ticks
SWEEP = [(256, 4), (1024, 1), (1024, 8), (2048, 4), (1024, 4)]  # BASELINE.json configs C2, C3', C4 (per node), C5, C3 (headline last)
# north_star's whole grid: "synthetic 256^2 - 2048^2 x {1, 4, 8}-cascade configs" (headline last)
SWEEP_GRID = [(n, c) for n in (256, 512, 1024, 2048) for c in (1, 4, 8) if (n, c) != (1024, 4)] + [(1024, 4)]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--map-size", type=int, default=1024)
    ap.add_argument("--cascades", type=int, default=None, help="cascades per GPU (default: 4 at --gpus 1 = the headline config; at --gpus N > 1 --total-cascades / N)")
    ap.add_argument("--total-cascades", type=int, default=None, help="cascades of the whole job, shared out evenly over the GPUs (strong scaling; default 8 at "
                                                                     "--gpus N > 1 = BASELINE config C4: 4 / 2 / 1 per GPU at N = 2 / 4 / 8)")
    ap.add_argument("--min-time", type=float, default=2.5, help="repeat the K-tick timed region until this many seconds have been timed (median reported)")
    ap.add_argument("--max-repeats", type=int, default=4000)
    ap.add_argument("--gather-every", type=int, default=-1, help="gather the maps every k ticks inside the timed region (0 = once, after it; "
                                                                   "-1 = auto: the smallest k whose gather hides under k ticks of compute)")
    ap.add_argument("--gather", choices=("all", "root"), default="root", help="gather to rank 0 (the consumer GPU), or all_gather to every rank")
    ap.add_argument("--no-overlap", action="store_true", help="serialise each gather with the compute stream (for comparison; default: side stream)")
    ap.add_argument("--prime-ms", type=float, default=600.0,
                    help="untimed clock priming before the W warm-up steps: at least this long, then until two consecutive 200-tick probes agree "
                         "within 1 %% (at most 4 s; 0 disables)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for rehearsing the "
                                                     "multi-rank control flow on a box with fewer GPUs than ranks)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses GPU 0 (numbers are meaningless)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single-stream", action="store_true", help="A/B: every context with OW_FLAG_SINGLE_STREAM (the tick-pair launches stay whole, on the one stream: no two chains)")
    ap.add_argument("--no-unmerged", action="store_true", help="time ow_run's regions only: no roofline.unmerged / update_all_calls / reference_schedule beside them")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU baseline sample size in seconds of host work (runs BEFORE the GPU regions, so that the "
                                                                    "GPU is busy for one contiguous stretch afterwards)")
    ap.add_argument("--no-measure-traffic", action="store_true", help="do not run the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) behind roofline.traffic; "
                                                                      "the figure of an earlier profiling visit (profiles/pmc_traffic.json) is quoted instead")
    ap.add_argument("--no-references", action="store_true", help="N > 1: skip per_gpu_alone / one_gpu_whole_job (the in-line scaling references)")
    ap.add_argument("--block-ms", type=float, default=40.0, help="N = 1: the timed regions of the four ways of driving the boundary alternate in blocks of about this long")
    ap.add_argument("--secondary-time", type=float, default=0.7, help="N = 1: seconds of timed regions for each of roofline.update_all_calls / unmerged / reference_schedule "
                                                                       "(and x 1.5 for each of roofline.other_configs)")
    ap.add_argument("--no-scene", action="store_true", help="skip roofline.scene_schedule (the reference scene's rate-limited, one-cascade-per-frame cadence)")
    ap.add_argument("--grid-time", type=float, default=0.5, help="N = 1: seconds of timed regions for each of the nine further grid configurations of roofline.other_configs")
    ap.add_argument("--no-grid", action="store_true", help="roofline.other_configs: BASELINE configs C5 and C2 only, not the rest of north_star's 256^2 .. 2048^2 x {1, 4, 8} grid")
    ap.add_argument("--no-other-configs", action="store_true", help="skip roofline.other_configs (2048^2 x 4 and 256^2 x 4 timed in the same process after the headline)")
    ap.add_argument("--sweep", action="store_true", help="one line per BASELINE configuration, appended to --sweep-out")
    ap.add_argument("--sweep-grid", action="store_true", help="like --sweep, over the whole grid 256^2 .. 2048^2 x {1, 4, 8} cascades")
    ap.add_argument("--sweep-out", default=os.path.join(ROOT, "profiles", "sweep.jsonl"))
    return ap.parse_args()


def cpu_baseline(n, cascades, seconds):
    """The oracle in reference-structure mode (separate modulate / table-driven radix-2 Stockham rows /
    transpose / rows / unpack passes = the reference's own algorithm) timed on this host's cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    from oracle import oracle as O
    from godotoceanwaves_amd.presets import UPDATE_DELTA
    O.build(native=True)
    g = H.oracle_generator(n, list(range(cascades)), native=True)
    g.update_all(UPDATE_DELTA)  # generates the spectra (excluded, like the GPU steady state)
    # bounded by TIME, not by a tick count estimated from one tick (a first tick of 25 ms promised 400 ticks in 10 s on a box whose 128 threads then sustained 128 ms per tick: 51 s)
    frames, t0 = 0, time.perf_counter()
    while frames < 400 and (frames < 3 or time.perf_counter() - t0 < seconds):
        g.update_all(UPDATE_DELTA)
        frames += 1
    dt = time.perf_counter() - t0
    cores = O.lib(True).owo_num_threads()
    g.close()
    out = {"value": round(frames * cascades / dt, 3), "unit": "maps/s", "cores": cores, "kind": "port",
           "sample": f"{frames} ticks of {n}^2 x {cascades} cascades (oracle, OpenMP x{cores}, {dt:.1f} s)"}
    try:
        out["reference_shaders"] = reference_shaders_baseline()
    except Exception as e:  # (the prebuilt oracle/_ref is test infrastructure: its absence costs this sub-object, nothing else)
        out["reference_shaders"] = {"error": str(e)[:160]}
    return out


def reference_shaders_baseline(n=256):
    """The reference's OWN compute shaders (oracle/_ref: its .glsl sources compiled as C++ through oracle/glsl_shim.h, workgroups run one after the
    other, barriers as fibres) on ONE host core: the steady-state dispatches of one update of one n^2 cascade -- spectrum_modulate, fft_compute,
    transpose, fft_compute, fft_unpack (wave_generator.gd:73-85).  An interpreter-grade execution of GPU code, reported because north_star asks for the
    reference's own compute path beside the port; a bounded sample (one 256^2 map: a 1024^2 map takes a minute this way)."""
    from oracle import ref as R
    from godotoceanwaves_amd.presets import UPDATE_DELTA, cascade_preset
    if not R.available():
        raise RuntimeError("oracle/_ref/libglsl_ref.so not built (needs the reference checkout at build time)")
    c = R.RefCascade(n, cascade_preset(0))
    c.update(UPDATE_DELTA)
    return {"value": round(1.0 / c.last_steady_s, 4), "unit": "maps/s", "cores": 1, "kind": "reference",
            "sample": f"1 update of one {n}^2 cascade through the reference's GLSL compiled as C++ (oracle/_ref, one thread, {c.last_steady_s:.1f} s; "
                      f"x {(1024 // n) ** 2} texels at 1024^2)"}


def measure_traffic(n, C, kernel, timeout_s=100, seamless=False):
    """Memory-side bytes per FULL launch of `kernel`, measured now: two rocprofv3 passes (FETCH_SIZE and WRITE_SIZE do not fit one: the TCC has
    four counter slots, MI355X_MICROARCH.md) over scripts/drive.py -- one ordinary tick, then 80 ticks through ow_run, i.e. the launches the timed
    region consists of -- with --kernel-trace beside the counters and nothing else.  FETCH_SIZE[KB] * 1024 * 2 (gfx950 tallies a 128-byte request
    as 64, same guide) + WRITE_SIZE[KB] * 1024, summed over the kernel's launches and divided by the full-launch equivalents (a run's first and
    last merged launch carry one pass each: launches - 1).  Returns (bytes_per_full_launch, detail) or raises."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    if any(k.startswith(("ROCPROF", "ROCPROFILER_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        raise RuntimeError("this process is itself being profiled: no nested rocprofv3")
    sums, launches = {}, {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="ow_pmc_", dir="/tmp")
        try:
            r = subprocess.run([exe, "--kernel-trace", "--pmc", ctr, "-d", d, "-o", "p", "--", sys.executable, os.path.join(ROOT, "scripts", "drive.py"),
                                "--map-size", str(n), "--cascades", str(C), "--frames", "81", "--warmup", "1"],
                               cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"}, capture_output=True, text=True, timeout=timeout_s)
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                raise RuntimeError(f"rocprofv3 --pmc {ctr} failed (rc {r.returncode}): {(r.stderr or r.stdout)[-200:]}")
            db = sqlite3.connect(dbs[0])
            rows = db.execute("select name, count(distinct dispatch_id), sum(counter_value) from pmc_events where counter_name = ? group by name", (ctr,)).fetchall()
            db.close()
            hit = [(nm, cnt, tot) for nm, cnt, tot in rows if f"::{kernel}<" in nm]
            if not hit:
                raise RuntimeError(f"{kernel} not among the profiled kernels: {[nm.split('(')[0][-40:] for nm, _, _ in rows]}")
            sums[ctr] = sum(t for _, _, t in hit)
            launches[ctr] = sum(c for _, c, _ in hit)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    # (ow_run's merged launches: a run's first and last launch carry one pass each = one full launch between them -- except the seamless stream
    #  of single-batch tick pairs, round 5, whose every launch carries both passes; other kernels: every launch is full)
    full = max(1, launches["FETCH_SIZE"] - 1) if (kernel.startswith("k_tick_") and not seamless) else max(1, launches["FETCH_SIZE"])
    nbytes = (sums["FETCH_SIZE"] * 2.0 + sums["WRITE_SIZE"]) * 1024.0 / full
    return int(round(nbytes)), {"FETCH_SIZE_KB_sum": round(sums["FETCH_SIZE"], 1), "WRITE_SIZE_KB_sum": round(sums["WRITE_SIZE"], 1),
                                "launches": launches["FETCH_SIZE"], "full_launch_equivalents": full}


def pmc_traffic(kernel, n, per_launch):
    """memory-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (NOT measured by this run)"""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        table = json.load(open(path))
    except Exception:
        return None
    key = f"{kernel}_{n}x{per_launch}"
    return table.get(key) if float(per_launch).is_integer() or isinstance(per_launch, int) else None


class Sensors:
    """sclk / mclk / socket power / temperatures of the device, one amdsmi call per sample (the bench line recorded no clock, so a
    slow region could not be told from a slow box).  Every failure is swallowed: the sensors must not cost the line."""

    def __init__(self, index=0):
        self.h, self.source, self.error = None, None, None
        try:
            import amdsmi
            self.smi = amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            self.h = hs[index if index < len(hs) else 0]
            self.source = "amdsmi_get_gpu_metrics_info"
            if self.sample() is None:
                self.h = None
        except Exception as e:  # noqa: BLE001
            self.h, self.error = None, f"{type(e).__name__}: {str(e)[:120]}"

    @staticmethod
    def _num(m, *keys):
        for k in keys:
            v = m.get(k)
            if isinstance(v, (list, tuple)):
                v = [x for x in v if isinstance(x, (int, float)) and 0 < x < 65535]
                v = sum(v) / len(v) if v else None
            if isinstance(v, (int, float)) and 0 <= v < 65535:
                return round(float(v), 1)
        return None

    def sample(self):
        if self.h is None:
            return None
        try:
            m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
            out = {"sclk_mhz": self._num(m, "current_gfxclks", "current_gfxclk", "average_gfxclk_frequency"),
                   "mclk_mhz": self._num(m, "current_uclk", "average_uclk_frequency"),
                   "power_w": self._num(m, "current_socket_power", "average_socket_power"),
                   "temp_hotspot_c": self._num(m, "temperature_hotspot", "temperature_edge"), "temp_mem_c": self._num(m, "temperature_mem")}
            return out if any(v is not None for v in out.values()) else None
        except Exception as e:  # noqa: BLE001
            self.error = f"{type(e).__name__}: {str(e)[:120]}"
            return None

    @staticmethod
    def summary(samples):
        """median per field over a list of samples"""
        samples = [x for x in samples if x]
        if not samples:
            return None
        out = {"samples": len(samples)}
        for k in samples[0]:
            v = [x[k] for x in samples if x.get(k) is not None]
            out[k] = round(statistics.median(v), 1) if v else None
        return out


class Driver:
    """One context driven through the C-ABI the way a C# / C host drives it: a record array that lives across calls (the library advances
    time and the foam rates in it, wave_generator.gd:101-106) and ONE foreign call per entry point -- no per-call packing of Python objects
    inside a timed region."""

    def __init__(self, gen, params):
        from godotoceanwaves_amd import _lib
        self.gen, self.lib, self.ctx, self.check = gen, gen._lib, gen.context, _lib.check
        self.C = len(params)
        self.arr = (_lib.ow_cascade_params * self.C)()
        for p, c in zip(params, self.arr):
            p._pack(c)

    def run(self, delta, frames):
        st = self.lib.ow_run(self.ctx, delta, self.arr, self.C, frames)
        if st:
            self.check(st)

    def update_all(self, delta):
        st = self.lib.ow_update_all(self.ctx, delta, self.arr, self.C)
        if st:
            self.check(st)

    def update(self, delta):
        st = self.lib.ow_update(self.ctx, delta, self.arr, self.C)
        if st:
            self.check(st)

    def process(self):
        st = self.lib.ow_process(self.ctx)
        if st:
            self.check(st)

    def sync(self):
        self.gen.sync()

    def hits(self):
        return self.gen.lookahead_stats()[0]

    def free(self):
        self.gen.free()


def scene_frames(drv, frame_hz, frames, jitter, ups=50.0, seed=12345, per_frame=None):
    """The reference scene's cadence, call by call: water.gd:75-82's rate limiter on a frame clock of `frame_hz` (each frame's delta scaled by
    1 +- `jitter`, a fixed pseudo-random sequence: real frame times are never equal), `ups` updates per second; every frame the generator node's
    _process drains ONE armed cascade (wave_generator.gd:56-63); what an update finds still armed it flushes (:94-98).
    per_frame(f): called around each frame's calls when given (returns a context manager-like pair of callables).  Returns the number of updates."""
    time_, next_update, updates = 0.0, 0.0, 0
    state = seed
    for f in range(frames):
        state = (state * 1103515245 + 12345) & 0x7FFFFFFF
        delta = (1.0 / frame_hz) * (1.0 + jitter * (state / 0x3FFFFFFF - 1.0))
        if per_frame:
            per_frame(f, True)
        if time_ >= next_update:                                    # water.gd:77
            target = 1.0 / (ups + 1e-10)                            # :78
            update_delta = target + (time_ - next_update)           # :79
            next_update = time_ + target                            # :80
            drv.update(update_delta)                                # :81 -> wave_generator.update
            updates += 1
        time_ += delta                                              # :82
        drv.process()                                               # the child node's _process, after the parent's
        if per_frame:
            per_frame(f, False)
    return updates


def measure_scene(torch, compute, drv, n, C, quick=False):
    """roofline.scene_schedule: see the module docstring.  Throughput form (nothing synchronises inside: us of GPU per update, maps/s) and latency
    form (the stream is idle at every frame, as in a scene that renders between the calls: GPU time of one frame's calls, p50 / p99)."""
    out = {"what": "water.gd:75-82 rate limiter, 50 updates/s, one ow_process per frame, leftovers flushed by the next ow_update",
           "bytes_per_texel": 72}
    for hz, jitter in ((144, 0.05), (60, 0.05), (144, 0.0)):
        frames = (300 if quick else 900) * (hz // 60 + 1) // 2
        scene_frames(drv, hz, max(60, frames // 5), jitter)          # warm-up (arms whatever the cadence arms)
        drv.sync()
        h0 = drv.hits()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        updates = scene_frames(drv, hz, frames, jitter, seed=777)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        hits = drv.hits() - h0
        # latency form: events around each frame's calls, the stream idle before
        ev, spans = {}, []

        def per_frame(f, begin):
            if begin:
                torch.cuda.synchronize()
                ev["a"], ev["b"] = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev["a"].record(compute)
            else:
                ev["b"].record(compute)
                ev["b"].synchronize()
                spans.append(ev["a"].elapsed_time(ev["b"]) * 1e3)
        scene_frames(drv, hz, 150 if quick else 400, jitter, seed=4242, per_frame=per_frame)
        spans.sort()
        key = f"{hz}hz_" + ("jitter5pct" if jitter else "fixed_clock")
        us_per_update = dt / max(1, updates) * 1e6
        out[key] = {"us_per_update": round(us_per_update, 2), "maps_per_s": round(updates * C / dt, 1), "updates": updates, "frames": frames,
                    "lookahead_hit_rate": round(hits / max(1, updates * C), 3),
                    "frac": round(72.0 * n * n * C / (us_per_update * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
                    "frame_gpu_us_p50": round(spans[len(spans) // 2], 1), "frame_gpu_us_p99": round(spans[min(len(spans) - 1, int(len(spans) * 0.99))], 1),
                    "frame_gpu_us_max": round(spans[-1], 1)}
    return out


# north_star's single-GPU reporting grid: 256^2 .. 2048^2 x {1, 4, 8}; (n, C) -> ticks per timed region (a region lasts 3 .. 50 ms)
GRID_STEPS = {(256, 1): 1000, (256, 4): 1000, (256, 8): 1000, (512, 1): 1000, (512, 4): 400, (512, 8): 400, (1024, 1): 400, (1024, 4): 200, (1024, 8): 100,
              (2048, 1): 200, (2048, 4): 200, (2048, 8): 100}


def measure_other_config(torch, compute, local_rank, n, C, steps, seconds, sensors, prime_s=0.3):
    """roofline.other_configs: one more configuration of north_star's grid in the same process -- its own context (created and freed here: one at a
    time), spectra generated, clocks primed for `prime_s`, then regions of EXACTLY `steps` ow_run ticks between two synchronisations, repeated for
    ~`seconds`, median reported; bytes per texel = the compact family's 72 (68.x in tick groups: foam stays in registers between a group's ticks)."""
    from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA
    gen = WaveGenerator()
    gen.map_size, gen.device_id, gen.stream = n, local_rank, compute.cuda_stream
    gen.single_stream = SINGLE_STREAM
    gen.init_gpu(max(2, C))
    drv = Driver(gen, [WaveCascadeParameters(**cascade_preset(i)) for i in range(C)])
    try:
        drv.update_all(UPDATE_DELTA)
        drv.run(UPDATE_DELTA, steps)
        torch.cuda.synchronize()
        tp = time.perf_counter()
        while time.perf_counter() - tp < prime_s:
            drv.run(UPDATE_DELTA, steps)
            torch.cuda.synchronize()
        samples, smp = [], []
        t_all = time.perf_counter()
        while time.perf_counter() - t_all < seconds or len(samples) < 3:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            drv.run(UPDATE_DELTA, steps)
            torch.cuda.synchronize()
            samples.append(time.perf_counter() - t0)
            if len(samples) % 8 == 1:
                smp.append(sensors.sample())
        tick = statistics.median(samples) / steps
        fam, depth = gen.last_kernel_family(), gen.tick_group_depth()
        kernel = MERGED_KERNEL.get(fam, fam) + ("_split" if (n == 2048 and fam == "tick_pairs_compact") else "")
        bpt = 72 - (4 * (depth - 1) / depth if fam == "tick_groups_compact" and depth > 1 else 0)  # (foam stays in registers between the ticks of a group)
        gbps = bpt * n * n * C / tick / 1e9
        clk = Sensors.summary(smp) or {}
        # (kept short: eleven of these ride in the one JSON line.  value = maps/s; frac = bytes_per_texel x texels / tick / 8 TB/s)
        return {"workload": f"{n}^2 x {C}", "steps_per_region": steps, "repeats": len(samples), "ms_per_step": round(tick * 1e3, 5),
                "value": round(C / tick, 1), "kernel": kernel, "bytes_per_texel": round(bpt, 2), "frac": round(gbps / HBM_PEAK_GBPS, 4),
                "clocks": {k: clk[k] for k in ("sclk_mhz", "power_w") if k in clk}}
    finally:
        drv.free()


def grid_scalars(other_configs, headline_frac):
    """north_star's 12-configuration grid as scalars of `roofline`: the fraction of 8 TB/s per configuration (headline included) in ONE short string,
    and the two other BASELINE configurations by name"""
    frac = {e["workload"]: e.get("frac") for e in other_configs}
    frac["1024^2 x 4"] = headline_frac
    fmt = lambda v: "-" if v is None else f"{v:.2f}".lstrip("0")
    by = {e["workload"]: e for e in other_configs}
    out = {"grid_frac_x1_x4_x8": "; ".join(f"{gn} " + "/".join(fmt(frac.get(f"{gn}^2 x {gc}")) for gc in (1, 4, 8)) for gn in (256, 512, 1024, 2048))}
    for key, wl in (("c5_2048x4", "2048^2 x 4"), ("c2_256x4", "256^2 x 4")):
        if wl in by and "frac" in by[wl]:
            out[key + "_frac"] = by[wl]["frac"]
            out[key + "_ms_per_step"] = by[wl]["ms_per_step"]
    return out


def measure(args, torch, dist, world, rank, local_rank, n, C, sensors):
    from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA
    from godotoceanwaves_amd import sharding

    layers = max(2, C)
    # The generator enqueues on a torch-owned stream and writes into torch-owned device memory, so that snapshots and RCCL
    # are ordered against it by ordinary stream semantics (PyTorch = memory + streams + collectives plumbing).
    compute = torch.cuda.Stream()
    disp = torch.zeros((layers, n, n, 4), dtype=torch.float16, device="cuda")
    norm = torch.zeros((layers, n, n, 4), dtype=torch.float16, device="cuda")
    torch.cuda.synchronize()

    def make_driver(**attrs):
        """a context on the shared stream and map arrays, with FRESH parameter objects: their dirty flags make THIS context generate its spectra
        (a context fed with consumed flags would run on all-zero spectra, and zeros run measurably FASTER: 52.0 against 55.4 us per tick at
        1024^2 x 4, less switching power, higher clock; round 3 fell for that once)"""
        g = WaveGenerator()
        g.map_size, g.device_id, g.stream = n, local_rank, compute.cuda_stream
        g.external_maps = (disp.data_ptr(), norm.data_ptr())
        g.single_stream = SINGLE_STREAM
        for k, v in attrs.items():
            setattr(g, k, v)
        g.init_gpu(layers)
        # global cascade ids: rank r owns cascades r*C .. r*C+C-1 (independent units; presets repeat with new seeds)
        return Driver(g, [WaveCascadeParameters(**cascade_preset(i)) for i in sharding.owned_cascades(rank, world, C)])

    main_drv = make_driver()
    gen = main_drv.gen
    gat = None
    if world > 1:
        gat = sharding.MapGatherer(torch, dist, world, rank, disp, norm, C, mode=args.gather, root=0,
                                   overlap=not args.no_overlap, compute_stream=compute)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    state = {"ticks": args.steps}  # ticks per timed region: K, or R x K under one outer pair of synchronisations (N > 1, see below)

    def region(drv, gather_every=0):
        """exactly state["ticks"] ticks, barrier + synchronize on both sides; returns max-over-ranks seconds"""
        ticks = state["ticks"]
        sync_all()
        t0 = time.perf_counter()
        if gat is not None and gather_every > 0:
            # a pipelined consumer: every chunk of k ticks BEGINS by shipping the maps as they stand (the previous chunk's last tick;
            # snapshot in stream order, bytes on the side stream), so each gather has its k ticks of compute to hide under
            done = 0
            while done < ticks:
                k = min(gather_every, ticks - done)
                gat.begin()
                drv.run(UPDATE_DELTA, k)
                done += k
            gat.wait()  # the last gather's bytes have arrived
        else:
            drv.run(UPDATE_DELTA, ticks)
        sync_all()
        return max_over_ranks(time.perf_counter() - t0)

    def timed(drv, gather_every, min_time):
        first = region(drv, gather_every)
        repeats = max(1, min(args.max_repeats, int(math.ceil(min_time / max(first, 1e-6)))))
        samples = [first] + [region(drv, gather_every) for _ in range(repeats - 1)]
        return statistics.median(samples), samples

    idle_clocks = sensors.sample()
    # ---- warm-up (includes the one-time spectrum generation) ----
    t_spec0 = time.perf_counter()
    main_drv.update_all(UPDATE_DELTA)
    main_drv.sync()
    spectrum_ms = (time.perf_counter() - t_spec0) * 1e3
    # the other ways of driving the boundary (N = 1), created and warmed BEFORE the priming: all timed regions then alternate on warm contexts
    others = {}
    other_errors = {}
    if world == 1 and not args.no_unmerged:
        ways = [("update_all_calls", {"run_as_calls": True}), ("unmerged", {"tick_groups": False}), ("reference_schedule", {"run_as_reference": True})]
        if not SINGLE_STREAM and ((n == 1024 and C in (4, 8)) or (n == 512 and C == 8)):
            # same-lease A/B of round 6's launch shape: the same ow_run regions with every tick-pair launch whole, on the one stream (OW_FLAG_SINGLE_STREAM)
            ways.append(("one_stream", {"single_stream": True}))
        for name, attrs in ways:
            try:  # secondary figures: whatever goes wrong here must not cost the headline line
                d = make_driver(**attrs)
                d.update_all(UPDATE_DELTA)
                d.run(UPDATE_DELTA, max(50, args.warmup))
                d.sync()
                others[name] = d
            except Exception as e:  # noqa: BLE001
                other_errors[name] = f"{type(e).__name__}: {e}"
    # ---- clock priming, untimed (not part of W or K): the chip's DVFS needs load to reach its steady state, and a region timed on the ramp -- or
    #      right after the 128-thread CPU leg -- is not the kernel's rate.  Until two consecutive 200-tick probes agree within 1 % (at least
    #      --prime-ms, at most 4 s) ----
    priming = {"ms": 0.0, "ticks": 0, "probes": 0, "stable": None, "first_probe_ms_per_step": None, "last_probe_ms_per_step": None}
    if args.prime_ms > 0:
        tp, prev = time.perf_counter(), None
        while True:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            main_drv.run(UPDATE_DELTA, 200)
            torch.cuda.synchronize()
            cur = (time.perf_counter() - t0) / 200
            priming["ticks"] += 200
            priming["probes"] += 1
            priming["first_probe_ms_per_step"] = priming["first_probe_ms_per_step"] or round(cur * 1e3, 5)
            priming["last_probe_ms_per_step"] = round(cur * 1e3, 5)
            el = (time.perf_counter() - tp) * 1e3
            stable = prev is not None and abs(cur - prev) <= 0.01 * prev
            prev = cur
            if (stable and el >= args.prime_ms) or el >= max(4000.0, args.prime_ms):
                priming["stable"] = bool(stable)
                break
        priming["ms"] = round((time.perf_counter() - tp) * 1e3, 1)
    if args.warmup > 1:
        main_drv.run(UPDATE_DELTA, args.warmup - 1)
    if gat is not None:
        gat.begin()
        gat.wait()

    # ---- gather cadence (N > 1): the smallest k whose gather hides under k ticks of compute, from two short probes ----
    gather_every, cadence = 0, None
    if world > 1:
        gather_every = args.gather_every
        if gather_every < 0:
            sync_all()
            t0 = time.perf_counter()
            main_drv.run(UPDATE_DELTA, 200)
            sync_all()
            tick_probe = max_over_ranks(time.perf_counter() - t0) / 200
            t0 = time.perf_counter()
            for _ in range(5):
                gat.begin()
                gat.wait()
            sync_all()
            gather_probe = max_over_ranks(time.perf_counter() - t0) / 5
            gather_every = max(1, int(math.ceil(1.25 * gather_probe / max(tick_probe, 1e-9))))  # 25 % slack: the links must never be the queue
            # the model the first real SCALE line can be checked against at a glance: a 1024^2 cascade's two maps are 16 MiB; the root's inbound
            # links carry one sender each (xGMI is point to point: ~153 GB/s per link, MI355X_MICROARCH.md)
            model_gather_ms = 16.0 * n * n * C / 153e9 * 1e3
            cadence = {"policy": "auto: smallest k with 1.25 x gather time <= k ticks",
                       "tick_probe_ms": round(tick_probe * 1e3, 5), "gather_probe_ms": round(gather_probe * 1e3, 4), "every_ticks": gather_every,
                       "model": {"gather_bytes_per_rank": 16 * n * n * C, "link_gbps_assumed": 153.0, "gather_ms_at_link_rate": round(model_gather_ms, 4),
                                 "predicted_every_ticks": max(1, int(math.ceil(1.25 * model_gather_ms / max(tick_probe * 1e3, 1e-9))))}}
        gather_every = max(0, gather_every)
        # A region of K = --steps ticks cannot hold a cadence longer than itself, and a region that ships ONE gather lasts max(K ticks, gather):
        # with the driver's --steps 20 that would time the links, not the pipeline.  So the region becomes R regions of K ticks back to back
        # under ONE outer barrier + synchronize pair, R the smallest count that gives the gather four full cadences: every figure of the line
        # (value, no_gather, gather_every_tick) is then timed over the same R x K ticks, and R is printed (`regions_per_sync`).
        if gather_every > 0 and args.steps < 4 * gather_every:
            state["ticks"] = args.steps * int(math.ceil(4 * gather_every / args.steps))

    # ---- references for the scaling question (N > 1), measured in this run, before the timed regions ----
    alone = whole_job = None
    if world > 1 and not args.no_references:
        # (i) every rank on its own: the same ticks timed locally -- no barrier, no gather, no process-group traffic inside the region
        torch.cuda.synchronize()
        mine = []
        for _ in range(5):
            t0 = time.perf_counter()
            main_drv.run(UPDATE_DELTA, max(200, args.steps))
            torch.cuda.synchronize()
            mine.append((time.perf_counter() - t0) / max(200, args.steps))
        rate = torch.tensor([C / statistics.median(mine)], dtype=torch.float64, device="cuda")
        rates = [torch.zeros_like(rate) for _ in range(world)]
        dist.all_gather(rates, rate)
        alone = [float(r.item()) for r in rates]
        # (ii) the whole job on ONE GPU: rank 0 runs all world x C cascades alone while the other ranks wait at the barrier -- the N = 1
        # point of the strong-scaling series on this very node (possible while the job is at most MAX_CASCADES = 8 cascades)
        total = world * C
        if total <= 8:
            sync_all()
            if rank == 0:
                try:
                    solo = WaveGenerator()
                    solo.map_size, solo.device_id = n, local_rank
                    solo.init_gpu(max(2, total))
                    sd = Driver(solo, [WaveCascadeParameters(**cascade_preset(g)) for g in range(total)])
                    sd.run(UPDATE_DELTA, 300)
                    sd.sync()
                    ts = []
                    for _ in range(5):
                        t0 = time.perf_counter()
                        sd.run(UPDATE_DELTA, 300)
                        sd.sync()
                        ts.append((time.perf_counter() - t0) / 300)
                    whole_job = {"cascades": total, "ms_per_step": round(statistics.median(ts) * 1e3, 5), "value": round(total / statistics.median(ts), 2),
                                 "unit": "maps/s", "launches": solo.last_kernel_family()}
                    solo.free()
                except Exception as e:  # noqa: BLE001  (a reference figure must not cost the line)
                    whole_job = {"error": f"{type(e).__name__}: {e}"}
            sync_all()
            main_drv.run(UPDATE_DELTA, 50)  # clocks back up on the ranks that waited

    # ---- timed regions ----
    block_clocks, hit_base, block_medians = {}, {}, []
    samples_of = {"run": []}
    if world == 1:
        # INTERLEAVED (the headline region was the slowest merged region of its own process): blocks of R regions each, ~--block-ms
        # long, alternate between the ways of driving the boundary -- run X run Y run Z ... -- until the headline has --min-time seconds of timed
        # regions and every other way --secondary-time.  A region is exactly K ticks between two synchronisations; every figure is the MEDIAN of
        # its regions over all its blocks (the first region of a block re-fetches its context's working set into the Infinity Cache).
        probe_region = region(main_drv)
        R = max(1, int(round(args.block_ms * 1e-3 / max(probe_region, 1e-6))))
        names = list(others)
        for nm in names:
            samples_of[nm] = []
            hit_base[nm] = others[nm].hits()
        order = []
        for nm in names or [None]:
            order += ["run"] + ([nm] if nm else [])
        budget = {"run": args.min_time, **{nm: min(args.min_time, args.secondary_time) for nm in names}}
        for _cycle in range(10000):
            for nm in list(order):
                if nm != "run" and nm not in others:
                    continue
                drv = main_drv if nm == "run" else others[nm]
                try:
                    blk = [region(drv) for _ in range(R)]
                except Exception as e:  # noqa: BLE001
                    if nm == "run":
                        raise
                    # a secondary figure must not cost the headline line: this way of driving the boundary drops out, with its error on the line
                    other_errors[nm] = f"{type(e).__name__}: {e}"
                    others.pop(nm).free()
                    samples_of.pop(nm, None)
                    budget.pop(nm, None)
                    continue
                samples_of[nm] += blk
                block_clocks.setdefault(nm, []).append(sensors.sample())
                if nm == "run":
                    block_medians.append(statistics.median(blk))
            if all(sum(samples_of[nm]) >= budget[nm] or len(samples_of[nm]) >= args.max_repeats for nm in samples_of):
                break
        samples = samples_of["run"]
        elapsed = statistics.median(samples)
    else:
        elapsed, samples = timed(main_drv, gather_every, args.min_time)
    group_depth = gen.tick_group_depth()
    launch_mode = gen.last_kernel_family()  # "tick_groups_compact": ow_run launched pass 2 of tick k with pass 1 of tick k + 1 (small batches)
    no_gather = every_tick = None
    if world > 1 and gather_every > 0:
        no_gather, _ = timed(main_drv, 0, args.min_time)
        every_tick = elapsed if gather_every == 1 else timed(main_drv, 1, args.min_time)[0]

    # ---- final gather (outside the timed region unless --gather-every) + sanity ----
    gather_ms = None
    if gat is not None:
        sync_all()
        g0 = time.perf_counter()
        gat.begin()
        gat.wait()
        gather_ms = max_over_ranks(time.perf_counter() - g0) * 1e3
        got = gat.maps()
        if got is not None:
            assert torch.equal(got[0][rank * C:(rank + 1) * C], disp[:C]) and bool(torch.isfinite(got[0].float()).all())
    assert bool(torch.isfinite(disp[:C].float()).all()) and float(disp[:C].float().abs().max()) > 0.0

    # ---- the other ways' figures (N = 1) ----
    ticks = state["ticks"]               # ticks per timed region (K, or R x K: see regions_per_sync)
    unmerged = unmerged_samples = calls = calls_hits = one_stream = None
    refsched = {}
    for nm, d in others.items():
        smp = samples_of[nm]
        hit_rate = (d.hits() - hit_base[nm]) / max(1, len(smp) * ticks * (C if nm == "reference_schedule" else 1))
        if nm == "unmerged":
            unmerged, unmerged_samples = statistics.median(smp), smp
        elif nm == "update_all_calls":
            calls, calls_hits = statistics.median(smp), hit_rate
        elif nm == "one_stream":
            one_stream = statistics.median(smp)
        else:
            refsched = {"seconds": statistics.median(smp), "hit_rate": hit_rate}
    unmerged_error, calls_error = other_errors.get("unmerged"), other_errors.get("update_all_calls")
    if "reference_schedule" in other_errors:
        refsched = {"error": other_errors["reference_schedule"]}
    unmerged_family = others["unmerged"].gen.last_kernel_family() if "unmerged" in others else None
    for d in others.values():
        d.free()

    # ---- per-kernel durations, in situ: during `probe` further ticks every launch carries start/stop HIP events bound
    #      to its own dispatch packet on the generator's stream (hipExtLaunchKernel): begin -> end of the kernel itself,
    #      the quantity a rocprofv3 kernel trace reports ----
    probe = 400  # (independent of --steps: a 50-tick probe right after an idle moment measured the clock ramp, 33 us where 400 ticks give 27.7)
    merged = launch_mode in MERGED_KERNEL
    # two chains (round 6): did the timed regions' tick-pair launches go out as two launches of half the cascades on two streams?  (asked BEFORE the
    # probes below: a launch that carries timing events stays whole)
    chains = 2 if (merged and gen.chain_stats() > 0) else 1
    gl_ms = gl_n = 0
    if merged:  # ow_run's merged launches (tick groups / tick pairs), each timed on its own
        gen.timing(2)
        main_drv.run(UPDATE_DELTA, probe)
        gen.sync()
        gl_ms, gl_n = gen.timing_read_launches()
        gen.timing(False)
    main_drv.run(UPDATE_DELTA, 200)  # untimed: clocks back up after the host-side bookkeeping above
    gen.timing(True)
    main_drv.run(UPDATE_DELTA, probe)
    gen.sync()
    p1_ms, p2_ms, launches = gen.timing_read()
    gen.timing(False)
    family = gen.last_kernel_family()
    assert unmerged_family in (None, family), (unmerged_family, family)
    # ---- the scene's real cadence (N = 1, headline run only) ----
    scene = None
    if world == 1 and not args.no_scene and not args.sweep:
        try:
            main_drv.run(UPDATE_DELTA, 100)
            scene = measure_scene(torch, compute, main_drv, n, C, quick=args.min_time < 0.5)
        except Exception as e:  # noqa: BLE001
            scene = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
    sync_all()
    main_drv.free()
    # ---- the rest of north_star's single-GPU grid in the same process (N = 1, headline run only): 256^2 .. 2048^2 x {1, 4, 8} minus the headline ----
    # BASELINE configs C5 (2048^2 x 4, the DRAM-bound one) and C2 (256^2 x 4, the 1000-frame loop) first and with the longer budget; the other nine
    # with ~0.5 s of regions each; one context at a time, created and freed inside measure_other_config
    other_configs = None
    if world == 1 and (n, C) == (1024, 4) and not args.no_other_configs and not args.sweep:
        other_configs = []
        del disp, norm
        torch.cuda.empty_cache()
        todo = [(2048, 4, args.secondary_time * 1.5, 0.3), (256, 4, args.secondary_time * 1.5, 0.3)]
        if not args.no_grid:
            todo += [(gn, gc, args.grid_time, 0.15) for gn in (256, 512, 1024, 2048) for gc in (1, 4, 8) if (gn, gc) not in ((1024, 4), (2048, 4), (256, 4))]
        for on, oc, osec, oprime in todo:
            try:
                other_configs.append(measure_other_config(torch, compute, local_rank, on, oc, GRID_STEPS[(on, oc)], osec, sensors, prime_s=oprime))
            except Exception as e:  # noqa: BLE001
                other_configs.append({"workload": f"{on}^2 x {oc}", "error": f"{type(e).__name__}: {str(e)[:200]}"})
        other_configs.sort(key=lambda e: [int(v) for v in e["workload"].replace("^2 x", "").split()])
    links = None
    if world > 1:
        # what lies between each rank's device and the consumer's, as the HIP runtime reports it (ow_query_link: peer access, link type, hops), so
        # that the line says by itself whether a rank's 16 B/texel went over xGMI (the model's 153 GB/s per link), PCIe, or a staged path
        try:
            from godotoceanwaves_amd import _lib as _L
            import ctypes as _C
            lk = _L.ow_group_link()
            _L.check(_L.load().ow_query_link(int(local_rank), 0, _C.byref(lk)))
            mine = {"rank": rank, **lk.as_dict()}
        except Exception as e:  # noqa: BLE001  (never the reason a scaling run fails)
            mine = {"rank": rank, "error": str(e)[:200]}
        links = [None] * world
        dist.all_gather_object(links, mine)
    if rank != 0:
        return None

    maps = ticks * C * world
    pairs_per_tick = launches / probe                   # the runtime may split a tick into several launch pairs (ow_runtime.hip batch_size)
    per_launch = C / pairs_per_tick                     # average cascades per launch (7 cascades go as 4 + 3 -> 3.5)
    per_launch = int(per_launch) if float(per_launch).is_integer() else round(per_launch, 3)
    texels = n * n * per_launch
    k1, k2 = FAMILY_BYTES[family]
    first = p1_ms >= p2_ms
    dom = ("k_pass1" if first else "k_pass2") + SUFFIX[family]  # the name rocprofv3 lists the kernel under
    if first and n == 2048 and family == "compact":
        dom = "k_pass1c_split"  # rows that span two waves: the split-plan pass 1
    dom_ms = max(p1_ms, p2_ms)
    dom_bpt, dom_contract = (k1, CONTRACT_BYTES[0]) if first else (k2, CONTRACT_BYTES[1])
    gbps = lambda nbytes, ms: nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    grouped = merged
    if grouped:
        # ow_run launched the timed ticks merged across ticks -- tick groups (k_tick_group_c_lp: pass 2 of GROUP ticks + pass 1 of the next
        # GROUP per launch) or tick pairs (k_tick_pair_c: pass 2 of one batch + pass 1 of the next, GROUP = 1): that kernel IS the timed
        # region -- K ticks are K * batches / GROUP launches of it back to back on one in-order stream (+ one: the two ends of a run carry
        # one pass each) -- so its average launch duration is the timed region divided by its launches, and that is the figure the
        # rocprofv3 kernel trace of the same command lists (closing visit of round 2: 52.48 us here, 52.41 us in the trace).
        # The same launches timed one by one with a pair of HIP events each (timing mode 2, `probe` further ticks) are reported beside it
        # (avg_launch_ms_events): those brackets include the event packets' own handling and scatter by +-5 % from visit to visit.
        # (p1 / p2 below are the one-launch-per-pass kernels of the same tick, probed after.)
        GROUP = max(1, group_depth)
        dom = MERGED_KERNEL[launch_mode] + ("_split" if (n == 2048 and launch_mode == "tick_pairs_compact") else "")  # (rows that span two waves)
        # (foam is read before the first and written after the last of a group's ticks only: 4 B/texel less for each tick in between)
        T = probe - 1
        groups = -(-T // GROUP)
        batches = max(1, round((gl_n - 1) / groups))   # tick pairs: a tick of more than 4 Mi texels is two batches, one launch each
        per_launch = C // batches if C % batches == 0 else round(C / batches, 3)
        dom_bpt, dom_contract, texels = (k1 + k2) * GROUP - 4 * (GROUP - 1), sum(CONTRACT_BYTES) * GROUP, n * n * C / batches
        dom_ms = elapsed / ticks * 1e3 * GROUP / batches          # time per tick x ticks per launch
        if chains == 2:
            # each launch of the timed regions went out as TWO launches of half its cascades on two streams, each stream a chain of back-to-back launches
            # of its own: a launch still lasts one tick (x ticks per launch / batches), two of them run at any time, each moves half the bytes
            per_launch = per_launch // 2 if isinstance(per_launch, int) and per_launch % 2 == 0 else per_launch / 2
            texels = texels / 2
        achieved = chains * gbps(dom_bpt * texels, dom_ms)
        contract = chains * gbps(dom_contract * texels, dom_ms)
        events_achieved = gbps(((k1 + k2) * T - 4 * (T - groups)) * n * n * C, gl_ms * gl_n)
    else:
        achieved = gbps(dom_bpt * texels, dom_ms)
        contract = gbps(dom_contract * texels, dom_ms)
    tick_s = elapsed / ticks
    tick_bpt = (dom_bpt / max(1, group_depth)) if grouped else (k1 + k2)
    tick_moved = tick_bpt * n * n * C / tick_s / 1e9          # per GPU
    tick_contract = sum(CONTRACT_BYTES) * n * n * C / tick_s / 1e9
    traffic = pmc_traffic(dom, n, per_launch)  # (per FULL launch of the merged kernels)
    # what the tick keeps alive between its uses, against the 256 MiB memory-side Infinity Cache: where it fits, the "HBM" rates below
    # (and TCC_EA-side PMC traffic) are partly cache hits, not DRAM traffic (MI355X_MICROARCH.md, HBM section)
    t_bpt = 32 if family in ("standard", "layer_parallel") else 20
    in_flight = 2 if launch_mode == "tick_pairs_compact" else (2 * max(1, group_depth) if launch_mode == "tick_groups_compact" else 1)
    batch_texels = n * n * (per_launch if not isinstance(per_launch, float) else math.ceil(per_launch))
    res = {"spectra_bytes": 12 * n * n * C, "intermediate_bytes": int(t_bpt * batch_texels * in_flight), "intermediate_batches_in_flight": in_flight,
           "foam_bytes": 2 * n * n * C, "maps_bytes": 16 * n * n * C, "infinity_cache_bytes": 256 << 20}
    res["reused_bytes"] = res["spectra_bytes"] + res["intermediate_bytes"] + res["foam_bytes"]
    res["reused_fits_infinity_cache"] = res["reused_bytes"] <= res["infinity_cache_bytes"]
    res["note"] = "where the re-read set (spectra, intermediate, foam) fits the Infinity Cache the rates are fabric-side (cache + DRAM) traffic"
    frac_of = lambda seconds: round(gbps((k1 + k2) * n * n * C, seconds / ticks * 1e3) / HBM_PEAK_GBPS, 4)
    clocks = None
    if world == 1:
        clocks = {"source": sensors.source, **({"error": sensors.error} if sensors.error and not sensors.source else {}),
                  "idle_before": idle_clocks, **{("run" if nm == "run" else nm): Sensors.summary(v) for nm, v in block_clocks.items()},
                  "sampled": "after every block of regions (~%g ms)" % args.block_ms}
    headline = (n, C) == (1024, 4)
    out = {
        "metric": "displacement+normal maps/sec, 1024^2 x 4 cascades; achieved HBM GB/s vs peak" if (headline and world == 1) else
                  (f"displacement+normal maps/sec, {n}^2 x {C} cascades; achieved HBM GB/s vs peak" if world == 1 else
                   f"displacement+normal maps/sec, {n}^2 x {C * world} cascades sharded {C} per GPU over {world} GPUs, maps gathered to the consumer GPU; "
                   f"achieved HBM GB/s vs peak (per GPU)"),
        "value": round(maps / elapsed, 2),
        "unit": "maps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(tick_s * 1e3, 5),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "repeats": len(samples),
        "ms_per_step_min_max": [round(min(samples) / ticks * 1e3, 5), round(max(samples) / ticks * 1e3, 5)],
        "timed_seconds": round(sum(samples), 4),
        "timed_region_s": round(sum(samples), 4),                      # GPU time inside the timed regions of `value` alone
        "timed_ticks_per_region": ticks, "regions_per_sync": ticks // args.steps,
        **({"interleaving": {"order": "run, " + ", run, ".join(others) if others else "run only", "regions_per_block": R, "blocks_of_value": len(block_medians),
                             "block_medians_ms_per_step": [round(b / ticks * 1e3, 5) for b in (block_medians if len(block_medians) <= 12 else
                                                                                                   block_medians[:6] + block_medians[-6:])],
                             "value_is": "the median region of ow_run over all its blocks"}} if world == 1 else {}),
        "config": {"workload": f"{n}^2 x {C} cascades per GPU, steady-state tick (modulate + 2-D IFFT + unpack/foam), "
                               f"delta=1/50 s, SURVEY 8d cascade table",
                   "map_size": n, "cascades_per_gpu": C, **({"single_stream": True} if SINGLE_STREAM else {}), "parallelism": f"cascade-sharded x{world}",
                   "launches": (f"tick groups: pass 2 of {group_depth} ticks and pass 1 of the next {group_depth} in one launch (k_tick_group_c_lp)"
                                if launch_mode == "tick_groups_compact" else
                                "tick pairs: pass 2 of one batch and pass 1 of the next in one launch (k_tick_pair_c)" +
                                (", as two chains: two launches of two cascades each on two streams" if chains > 1 else "")) if grouped
                               else "one pair of launches per batch and tick",
                   "gather": (f"{args.gather}, every {gather_every} ticks (timed), " + ("serialised" if args.no_overlap else "snapshot + side stream"))
                             if (world > 1 and gather_every) else (f"{args.gather}, final, untimed" if world > 1 else "none"),
                   **({"rehearsal": f"backend={args.backend}, share_gpu={args.share_gpu}: NOT a measurement"}
                      if (args.share_gpu or (world > 1 and args.backend != "nccl")) else {})},
        "roofline": {
            "bound": "hbm", "kernel": dom, "kernel_family": family,
            # bytes this kernel must move (its family's design bytes) / its average launch duration
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "traffic": traffic,
            # (the driver's record keeps the first SCALAR keys of this object: the grid's fractions as scalars here, the entries themselves -- a list --
            #  at the end of the line, where a record that keeps the tail of stdout still has them)
            **(grid_scalars(other_configs, round(achieved / HBM_PEAK_GBPS, 4)) if other_configs is not None else {}),
            **({"scene_schedule": scene} if scene is not None else {}),
            **({"clocks": clocks} if clocks is not None else {}),
            **({"one_stream": {"ms_per_step": round(one_stream / ticks * 1e3, 5), "value": round(maps / one_stream, 2), "unit": "maps/s", "frac": frac_of(one_stream),
                               "launches": "ow_run as in `value`, every tick-pair launch whole on the one stream (OW_FLAG_SINGLE_STREAM): the same-lease A/B of the two chains"}}
               if one_stream is not None else ({"one_stream": {"error": other_errors["one_stream"]}} if "one_stream" in other_errors else {})),
            **({"update_all_calls": {"ms_per_step": round(calls / ticks * 1e3, 5), "value": round(maps / calls, 2), "unit": "maps/s",
                                     "lookahead_hit_rate": round(calls_hits, 4), "frac": frac_of(calls),
                                     "launches": "one ow_update_all per tick (OW_FLAG_RUN_AS_CALLS), its adaptive look-ahead on"}}
               if calls is not None else ({"update_all_calls": {"error": calls_error}} if calls_error else {})),
            **({"unmerged": {"ms_per_step": round(unmerged / ticks * 1e3, 5), "value": round(maps / unmerged, 2), "unit": "maps/s",
                             "frac": frac_of(unmerged),
                             "launches": "one launch per pass and batch (OW_FLAG_NO_TICK_GROUPS)",
                             "ms_per_step_min_max": [round(min(unmerged_samples) / ticks * 1e3, 5), round(max(unmerged_samples) / ticks * 1e3, 5)],
                             "bytes_per_texel": k1 + k2, "achieved": round(gbps((k1 + k2) * n * n * C, unmerged / ticks * 1e3), 1),
                             "frac_of_copy_ceiling": round(gbps((k1 + k2) * n * n * C, unmerged / ticks * 1e3) / COPY_CEILING_GBPS, 4),
                             "kernels": {("k_pass1" + SUFFIX[family] if not (n == 2048 and family == "compact") else "k_pass1c_split"):
                                             {"avg_ms_events": round(p1_ms, 5), "frac": round(gbps(k1 * n * n * (C / pairs_per_tick), p1_ms) / HBM_PEAK_GBPS, 4)},
                                         "k_pass2" + SUFFIX[family]:
                                             {"avg_ms_events": round(p2_ms, 5), "frac": round(gbps(k2 * n * n * (C / pairs_per_tick), p2_ms) / HBM_PEAK_GBPS, 4)}}}}
               if unmerged is not None else ({"unmerged": {"error": unmerged_error}} if unmerged_error else {})),
            **({"reference_schedule": ({"ms_per_step": round(refsched["seconds"] / ticks * 1e3, 5), "value": round(maps / refsched["seconds"], 2), "unit": "maps/s",
                                         "lookahead_hit_rate": round(refsched["hit_rate"], 4), "frac": frac_of(refsched["seconds"]),
                                         "cadence": "regular (every update the same delta); the scene's own cadence is scene_schedule",
                                         "launches": "per tick one ow_update + one ow_process per cascade (OW_FLAG_RUN_AS_REFERENCE_SCHEDULE)"}
                                        if "seconds" in refsched else refsched)} if refsched else {}),
            "bytes_per_texel": dom_bpt, "bytes_per_launch": int(dom_bpt * texels),
            "bytes_basis": "design bytes of the launched kernel family (DESIGN.md section 3)" +
                           (f"; one launch = both passes of {max(1, group_depth)} tick(s) of {per_launch} cascade(s), duration = timed region / launches" if grouped else "") +
                           (f" of ONE of the {chains} streams: {chains} such launches run at any time (two chains), achieved = {chains} x bytes_per_launch / avg_launch_ms" if chains > 1 else ""),
            **({"concurrent_launches": chains} if chains > 1 else {}),
            "frac_of_copy_ceiling": round(achieved / COPY_CEILING_GBPS, 4), "copy_ceiling": COPY_CEILING_GBPS,
            # SURVEY 8d's contract bytes (four-layer FP32 intermediate, 104 B/texel per map) over the same duration: a
            # figure of merit against a design that moves more, NOT a bandwidth (it can exceed the copy ceiling)
            "contract_bytes_per_texel": dom_contract, "contract_gbps": round(contract, 1), "frac_contract_104": round(contract / HBM_PEAK_GBPS, 4),
            "traffic_source": "profiles/pmc_traffic.json: rocprofv3 --pmc passes of an earlier visit, NOT measured by this run" if traffic else None,
            "traffic_gbps": round(gbps(traffic, dom_ms), 1) if traffic else None,
            "avg_launch_ms": round(dom_ms, 5), "pass1_ms": round(p1_ms, 5), "pass2_ms": round(p2_ms, 5),
            **({"avg_launch_ms_events": round(gl_ms, 5), "achieved_events": round(events_achieved, 1), "launches_timed_events": gl_n,
                **({"events_note": "a launch that carries timing events stays whole, on one stream: these are the one-stream launches of all the cascades"} if chains > 1 else {})} if grouped else {}),
            "launches_timed": launches, "launch_pairs_per_tick": round(pairs_per_tick, 3), "cascades_per_launch": per_launch,
            "residency": res,
            **({"other_configs": other_configs} if other_configs is not None else {}),
            "tick": {"bytes_per_texel": tick_bpt, "achieved": round(tick_moved, 1), "frac": round(tick_moved / HBM_PEAK_GBPS, 4),
                     "frac_of_copy_ceiling": round(tick_moved / COPY_CEILING_GBPS, 4),
                     "contract_gbps": round(tick_contract, 1), "frac_contract_104": round(tick_contract / HBM_PEAK_GBPS, 4)},
        },
        "frames_per_s": round(ticks * world / elapsed, 2),
        "spectrum_init_ms": round(spectrum_ms, 3),
        "clock_priming": priming,
    }
    if gat is not None:
        out["links_to_root"] = links
        out["final_gather_ms"] = round(gather_ms, 3)
        out["gather_bytes"] = {"sent_per_rank": gat.bytes_sent, "received_rank0": gat.bytes_received}
        out["gather"] = {"mode": gat.mode, **({"fallback": gat.fallback} if gat.fallback else {}), "every_ticks": gather_every, "overlap": not args.no_overlap, **(cadence or {}),
                         "regions_per_sync": ticks // args.steps,
                         "bound": "compute, if value == no_gather.value within noise: every gather hides under its chunk of ticks (the cadence was chosen for that); "
                                  "links otherwise",
                         # what the links deliver: bytes into the consumer per gather / the time between gathers in the timed region
                         "gathers_per_s": round(ticks / max(1, gather_every) / elapsed, 2) if gather_every else None,
                         "root_inbound_gbps": round(gat.bytes_received * (world - 1) / world * (ticks / max(1, gather_every)) / elapsed / 1e9, 2)
                                              if gather_every else None}
    if no_gather is not None:
        out["no_gather"] = {"ms_per_step": round(no_gather / ticks * 1e3, 5), "value": round(maps / no_gather, 2)}
    if world > 1:
        out["headline_basis"] = (f"value = {ticks} ticks ({ticks // args.steps} region(s) of --steps {args.steps} under one barrier + synchronize pair) with the maps "
                                 f"gathered to rank 0 every {gather_every} ticks on a side stream; no_gather = the same ticks without the exchange"
                                 if gather_every else "value = the timed ticks without any exchange (--gather-every 0): one final gather after the region, untimed")
    if alone:
        mean_alone = sum(alone) / len(alone)
        out["per_gpu_alone"] = {"value": round(mean_alone, 2), "unit": "maps/s per GPU", "min": round(min(alone), 2), "max": round(max(alone), 2),
                                "what": f"each rank's own rate on {n}^2 x {C}, timed locally: no barrier, no gather", "sum_over_gpus": round(sum(alone), 2)}
        out["speedup_with_gather"] = round(out["value"] / mean_alone, 3)        # in units of one GPU on the per-GPU config: ideal = n_gpus
        if no_gather is not None:
            out["speedup_no_gather"] = round(maps / no_gather / mean_alone, 3)
        out["parallel_efficiency"] = {"with_gather": round(out["value"] / sum(alone), 4),
                                      **({"no_gather": round(maps / no_gather / sum(alone), 4)} if no_gather is not None else {})}
        # what the first real SCALE line should look like if nothing but the per-GPU kernels bounds it (cascades are independent: no data-path
        # collective; the gather hides under its chunk of ticks when the cadence holds): the model next to the measurement
        out["expected"] = {"model": "value = sum of the ranks' own rates (per_gpu_alone.sum_over_gpus); the gather costs nothing while every_ticks >= gather.model.predicted_every_ticks",
                           "value_if_compute_bound": round(sum(alone), 2), "measured_over_expected": round(out["value"] / max(1e-9, sum(alone)), 4)}
    if whole_job:
        out["one_gpu_whole_job"] = {**whole_job, "what": f"rank 0 running all {world * C} cascades alone (ow_run), the other ranks idle: the N = 1 point of the "
                                                          "strong-scaling series, measured in this run"}
        if whole_job.get("value"):
            out["speedup_vs_one_gpu_with_gather"] = round(out["value"] / whole_job["value"], 3)   # north_star: >= 6 at 8 GPUs
            if no_gather is not None:
                out["speedup_vs_one_gpu_no_gather"] = round(maps / no_gather / whole_job["value"], 3)
    if every_tick is not None:
        out["gather_every_tick"] = {"ms_per_step": round(every_tick / ticks * 1e3, 5), "value": round(maps / every_tick, 2),
                                    "root_inbound_gbps": round(gat.bytes_received * (world - 1) / world * ticks / every_tick / 1e9, 2)}
    return out


def main():
    global SINGLE_STREAM
    args = parse()
    SINGLE_STREAM = bool(args.single_stream)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)

    # N = 1: the headline config (C3, 1024^2 x 4).  N > 1: BASELINE config C4 as a strong-scaling series -- 8 cascades shared out evenly, the same
    # job at every N (4 / 2 / 1 per GPU at N = 2 / 4 / 8); --cascades C pins C per GPU instead (weak scaling)
    args.scaling = "weak"
    if args.cascades is None:
        if world == 1 and args.total_cascades is None:
            args.cascades = 4
        else:
            total = args.total_cascades if args.total_cascades is not None else 8
            if total % world != 0 or total < world:
                raise SystemExit(f"--total-cascades {total} does not share out evenly over {world} GPUs")
            args.cascades = total // world
            args.scaling = "strong" if world > 1 else "weak"
    if args.sweep_grid:
        args.sweep = True
    configs = SWEEP_GRID if args.sweep_grid else (SWEEP if args.sweep else [(args.map_size, args.cascades)])
    sensors = Sensors(local_rank)
    for n, C in configs:
        # The CPU leg runs FIRST (rank 0, N = 1): everything after it is GPU work in one contiguous stretch -- clock priming, warm-up, the
        # timed regions, the kernel probes, the one-launch-per-pass region -- so that a monitor sampling the device every few seconds sees it
        # busy (round 3's run had its < 3 s of GPU time in front of 15 s of host-only baseline and was sampled as idle four times out of four)
        cpu = None
        if rank == 0 and not args.no_cpu_baseline and world == 1:
            try:  # (the CPU leg must not cost the line; the oracle is test infrastructure and may be absent from a deployment)
                cpu = cpu_baseline(n, C, args.cpu_seconds if not args.sweep else min(args.cpu_seconds, 8.0))
            except Exception as e:  # noqa: BLE001
                cpu = {"value": None, "unit": "maps/s", "cores": 0, "kind": "port", "sample": f"failed: {type(e).__name__}: {e}"}
        t_gpu0 = time.perf_counter()
        out = measure(args, torch, dist, world, rank, local_rank, n, C, sensors)
        if rank == 0:
            out["gpu_phase_s"] = round(time.perf_counter() - t_gpu0, 2)   # contiguous GPU activity of this configuration (priming .. last region)
            if world == 1 and not args.no_measure_traffic and not args.sweep:
                # roofline.traffic measured by THIS run (after the timed work: the profiled child process shares the GPU with nothing)
                rf = out["roofline"]
                try:
                    got, detail = measure_traffic(n, C, rf["kernel"], seamless=rf["kernel"].startswith("k_tick_pair_c") and rf["cascades_per_launch"] * rf.get("concurrent_launches", 1) == C)
                    rf["traffic"] = got
                    # (FETCH_SIZE[KB] * 1024 * 2 (gfx950) + WRITE_SIZE[KB] * 1024 per full launch -- every pair launch of a single-batch run carries both passes --
                    #  between the XCD L2s and the fabric, Infinity-Cache hits included: DESIGN.md section 6)
                    rf["traffic_source"] = "measured by this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two passes, 81 ticks through ow_run), gfx950 correction applied"
                    rf["traffic_detail"] = detail
                    rf["traffic_bytes_per_texel"] = round(got / max(1, rf["bytes_per_launch"]) * rf["bytes_per_texel"], 2)
                    rf["traffic_gbps"] = round(rf.get("concurrent_launches", 1) * got / (rf["avg_launch_ms"] * 1e-3) / 1e9, 1)
                except Exception as e:  # noqa: BLE001  (the profiling leg must not cost the line: the earlier visit's figure stays, labelled)
                    rf["traffic_measurement_failed"] = f"{type(e).__name__}: {str(e)[:200]}"
            if cpu is not None:
                out["cpu_baseline"] = cpu
                if cpu.get("value"):
                    out["gpu_vs_cpu"] = round(out["value"] / cpu["value"], 1)
            line = json.dumps(out)
            if args.sweep:
                os.makedirs(os.path.dirname(args.sweep_out), exist_ok=True)
                with open(args.sweep_out, "a") as f:
                    f.write(line + "\n")
            print(line, flush=True)
        if world > 1:
            dist.barrier()

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
