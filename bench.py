#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its headline config, one process per GPU.

  metric : displacement+normal maps/sec, 1024^2 x 4 cascades (1 "map" = one cascade update = one
           displacement layer + one normal/foam layer); achieved HBM GB/s vs peak in `roofline`
  step   : one simulation tick = time-modulate + 2-D IFFT + unpack/foam of all 4 cascades of this rank
           (steady state: the spectra h0 / omega are already resident in HBM; spectrum generation runs once
           during warm-up, like the reference's should_generate_spectrum path, and is reported separately)
  N > 1  : cascades/tiles are independent units: every rank owns its own 4 cascades
           (weak scaling, no data-path collective); one RCCL all_gather of the finished maps runs AFTER the
           timed region ("final gather"), or every k ticks inside it with --gather-every k.

Launch:  python bench.py [--gpus 1] [--steps K] [--warmup W]
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
                bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md:35
# algorithmic bytes per texel per cascade-update: pass 1 reads h0 (16) and writes the
# FP32 intermediate (32); pass 2 reads it (32), reads the previous normal texel for foam (8) and writes
# the two RGBA16F maps (8 + 8).  (The 4 B/texel omega plane pass 1 also reads is NOT counted.)
BYTES_PASS1, BYTES_PASS2 = 48, 56
BYTES_MAP = BYTES_PASS1 + BYTES_PASS2  # 104


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--map-size", type=int, default=1024)
    ap.add_argument("--cascades", type=int, default=4, help="cascades per GPU")
    ap.add_argument("--gather-every", type=int, default=0, help="RCCL all_gather of the maps every k ticks inside the timed region (0 = once, after it)")
    ap.add_argument("--prime-ms", type=float, default=300.0,
                    help="untimed clock priming before the W warm-up steps: the chip's DVFS needs tens of ms of load to reach its "
                         "steady clock, and a short run would otherwise time the ramp (0 disables)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for rehearsing the "
                                                     "multi-rank control flow on a box with fewer GPUs than ranks)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses GPU 0 (numbers are meaningless)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=0, help="CPU baseline sample size in ticks (0 = auto, ~10-20 s)")
    return ap.parse_args()


def cpu_baseline(n, cascades):
    """The oracle in reference-structure mode (separate modulate / table-driven radix-2 Stockham rows /
    transpose / rows / unpack passes = the reference's own algorithm) timed on this host's cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    from oracle import oracle as O
    from godotoceanwaves_amd.presets import UPDATE_DELTA
    O.build(native=True)
    g = H.oracle_generator(n, list(range(cascades)), native=True)
    g.update_all(UPDATE_DELTA)  # generates the spectra (excluded, like the GPU steady state)
    t0 = time.perf_counter()
    g.update_all(UPDATE_DELTA)
    one = time.perf_counter() - t0
    frames = max(1, min(200, int(12.0 / max(one, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(frames):
        g.update_all(UPDATE_DELTA)
    dt = time.perf_counter() - t0
    cores = O.lib(True).owo_num_threads()
    g.close()
    return {"value": round(frames * cascades / dt, 3), "unit": "maps/s", "cores": cores, "kind": "port",
            "sample": f"{frames} ticks of {n}^2 x {cascades} cascades (oracle, OpenMP x{cores}, {dt:.1f} s)"}


def main():
    args = parse()
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

    from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA
    from godotoceanwaves_amd import sharding

    n, C = args.map_size, args.cascades
    layers = max(2, C)
    # outputs live in torch-owned device memory so RCCL can gather them (PyTorch = memory + collectives plumbing)
    disp = torch.zeros((layers, n, n, 4), dtype=torch.float16, device="cuda")
    norm = torch.zeros((layers, n, n, 4), dtype=torch.float16, device="cuda")
    gen = WaveGenerator()
    gen.map_size = n
    gen.device_id = local_rank
    gen.external_maps = (disp.data_ptr(), norm.data_ptr())
    gen.init_gpu(layers)
    # global cascade ids: rank r owns cascades r*C .. r*C+C-1 (independent units; presets repeat with new seeds)
    params = [WaveCascadeParameters(**cascade_preset(g)) for g in sharding.owned_cascades(rank, world, C)]

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    gathered = None
    if world > 1:
        gathered = sharding.alloc_gather_buffers(torch, world, disp, norm)

    def gather():
        gen.sync()
        sharding.gather_maps(dist, gathered, disp, norm)

    # ---- warm-up (includes the one-time spectrum generation) ----
    t_spec0 = time.perf_counter()
    gen.update_all(UPDATE_DELTA, params)
    gen.sync()
    spectrum_ms = (time.perf_counter() - t_spec0) * 1e3
    prime_ticks = 0
    if args.prime_ms > 0:  # untimed: bring the clocks to their steady state (not part of W or K)
        tp = time.perf_counter()
        while (time.perf_counter() - tp) * 1e3 < args.prime_ms:
            gen.run(UPDATE_DELTA, params, 50)
            gen.sync()
            prime_ticks += 50
    if args.warmup > 1:
        gen.run(UPDATE_DELTA, params, args.warmup - 1)
    if world > 1:
        gather()
    sync_all()

    # ---- timed region: exactly K ticks ----
    t0 = time.perf_counter()
    if world > 1 and args.gather_every > 0:
        done = 0
        while done < args.steps:
            k = min(args.gather_every, args.steps - done)
            gen.run(UPDATE_DELTA, params, k)
            gather()
            done += k
    else:
        gen.run(UPDATE_DELTA, params, args.steps)
    sync_all()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # ---- final gather (outside the timed region unless --gather-every) + sanity ----
    gather_ms = None
    if world > 1:
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        gather()
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
        assert torch.equal(gathered[0][rank], disp) and bool(torch.isfinite(gathered[0].float()).all())
    assert bool(torch.isfinite(disp.float()).all()) and float(disp.float().abs().max()) > 0.0

    # ---- per-kernel durations, in situ: during `probe` further ticks every launch carries start/stop HIP events bound
    #      to its own dispatch packet on the generator's stream (hipExtLaunchKernel): begin -> end of the kernel itself,
    #      the quantity a rocprofv3 kernel trace reports ----
    gen.timing(True)
    probe = max(50, min(400, args.steps))
    gen.run(UPDATE_DELTA, params, probe)
    gen.sync()
    p1_ms, p2_ms, launches = gen.timing_read()
    gen.timing(False)
    family = gen.last_kernel_family()
    suffix = {"standard": "", "layer_parallel": "_lp", "compact": "c", "layer_parallel_compact": "c_lp"}[family]
    per_launch = gen.last_batch_cascades()  # cascades per pair of launches (the runtime may split a tick: ow_runtime.hip batch_size)
    sync_all()

    if rank == 0:
        maps = args.steps * C * world
        texels = n * n * per_launch  # per launch (the runtime batches cascades so that T stays in the Infinity Cache)
        dom = ("k_pass1" if p1_ms >= p2_ms else "k_pass2") + suffix  # the name rocprofv3 lists the kernel under
        dom_ms = max(p1_ms, p2_ms)
        dom_bytes = (BYTES_PASS1 if p1_ms >= p2_ms else BYTES_PASS2) * texels
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        frame_gbps = BYTES_MAP * n * n * maps / elapsed / 1e9 / world  # per-GPU algorithmic GB/s over the whole tick
        traffic = None
        prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(prof):
            try:
                traffic = json.load(open(prof)).get(f"{dom}_{n}x{per_launch}")
            except Exception:
                traffic = None
        out = {
            "metric": "displacement+normal maps/sec, 1024^2 x 4 cascades; achieved HBM GB/s vs peak",
            "value": round(maps / elapsed, 2),
            "unit": "maps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{n}^2 x {C} cascades per GPU, steady-state tick (modulate + 2-D IFFT + unpack/foam), "
                                   f"delta=1/50 s, SURVEY 8d cascade table",
                       "map_size": n, "cascades_per_gpu": C, "parallelism": f"cascade-sharded x{world}",
                       "gather": ("every %d ticks (timed)" % args.gather_every) if (world > 1 and args.gather_every) else
                                 ("final, untimed" if world > 1 else "none"),
                       **({"rehearsal": f"backend={args.backend}, share_gpu={args.share_gpu}: NOT a measurement"}
                          if (args.share_gpu or (world > 1 and args.backend != "nccl")) else {})},
            "roofline": {"bound": "hbm", "kernel": dom, "kernel_family": family, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         # what really crossed the memory interface per second (PMC bytes / this run's launch time): the compact
                         # kernels move fewer bytes than SURVEY 8d's algorithmic 104 B/texel, so `achieved` can exceed it
                         "traffic_gbps": round(traffic / (dom_ms * 1e-3) / 1e9, 1) if (traffic and dom_ms > 0) else None,
                         "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": round(dom_ms, 5),
                         "pass1_ms": round(p1_ms, 5), "pass2_ms": round(p2_ms, 5), "launches_timed": launches,
                         "cascades_per_launch": per_launch,
                         "tick_achieved_gbps_per_gpu": round(frame_gbps, 1), "tick_frac": round(frame_gbps / HBM_PEAK_GBPS, 4)},
            "frames_per_s": round(args.steps * world / elapsed, 2),
            "spectrum_init_ms": round(spectrum_ms, 3),
            "clock_priming": {"ms": args.prime_ms, "ticks": prime_ticks},
        }
        if gather_ms is not None:
            out["final_gather_ms"] = round(gather_ms, 3)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(n, C)
            out["gpu_vs_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
