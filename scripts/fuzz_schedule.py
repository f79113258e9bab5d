#!/usr/bin/env python3
"""Random call schedules against the boundary: a context with every merged launch shape on (tick groups, tick pairs, the look-ahead of
ow_update_all / ow_process) and a context that never merges anything (OW_FLAG_NO_TICK_GROUPS) are driven through the same random sequence of
calls -- update_all with repeating and changing deltas, update + some or all of its process calls (leftovers flushed by the next update),
runs of a few ticks, live edits (tile length: dirty; whitecap: not), fewer cascades, a restored foam plane, readbacks in between -- and must
hold the same bits in every map at every check and at the end.
    python scripts/fuzz_schedule.py [schedules per configuration [first seed]] [--caller-stream]"""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA  # noqa: E402

# every launch shape of ow_run: tick groups (256^2, 512^2 x 2, 1024^2 x 1), the seamless single-batch pair stream (512^2 x 8, 1024^2 x 3, the headline
# 1024^2 x 4, 2048^2 x 1) and the multi-batch cascade-major stream (1024^2 x 8 = 4 + 4, 2048^2 x 2 = 1 + 1), whose 64-tick blocks the long runs below cross
CONFIGS = [(256, 4), (256, 1), (512, 2), (512, 8), (1024, 1), (1024, 3), (256, 8), (2048, 1), (1024, 4), (1024, 8), (2048, 2)]
DELTAS = (UPDATE_DELTA, UPDATE_DELTA, UPDATE_DELTA, UPDATE_DELTA, 0.03, 1.0 / 60.0)


CALLER_STREAM = False   # --caller-stream: the merging context runs on a stream and map arrays of the CALLER's (torch's), and the maps are read by work enqueued on that
                        # stream with NO synchronising call of the library in between -- the ordering a second chain must be joined for before every call returns


def make(n, count, merge):
    gen = WaveGenerator()
    gen.map_size, gen.tick_groups = n, merge
    if merge and CALLER_STREAM:
        import torch
        layers = max(2, count)
        gen._st = torch.cuda.Stream()
        gen._disp = torch.zeros((layers, n, n, 4), dtype=torch.float16, device="cuda")
        gen._norm = torch.zeros_like(gen._disp)
        torch.cuda.synchronize()
        gen.stream, gen.external_maps = gen._st.cuda_stream, (gen._disp.data_ptr(), gen._norm.data_ptr())
    gen.init_gpu(max(2, count))
    return gen, [WaveCascadeParameters(**cascade_preset(i)) for i in range(count)]


def compare(a, b, count, where):
    if hasattr(a, "_st"):
        import torch
        with torch.cuda.stream(a._st):          # ordered behind the library's work by the caller's stream alone
            d_dev, n_dev = a._disp.clone(), a._norm.clone()
        a._st.synchronize(); b.sync()
        for i in range(count):
            db, nb = b.get_maps(i)
            if not (np.array_equal(d_dev[i].cpu().numpy().view(np.uint16), db.view(np.uint16)) and np.array_equal(n_dev[i].cpu().numpy().view(np.uint16), nb.view(np.uint16))):
                raise AssertionError(f"maps of cascade {i} differ {where} (read on the caller's stream)")
        return
    a.sync(); b.sync()
    for i in range(count):
        da, na = a.get_maps(i)
        db, nb = b.get_maps(i)
        if not (np.array_equal(da.view(np.uint16), db.view(np.uint16)) and np.array_equal(na.view(np.uint16), nb.view(np.uint16))):
            raise AssertionError(f"maps of cascade {i} differ {where}")


def schedule(n, count, seed, ops=60):
    """returns (calls issued, calls served from work computed ahead)"""
    rng = random.Random(seed)
    a, pa = make(n, count, True)
    b, pb = make(n, count, False)
    log = []

    def both(name, f):
        log.append(name)
        f(a, pa); f(b, pb)

    delta = UPDATE_DELTA
    try:
        for step in range(ops):
            if rng.random() < 0.25:
                delta = rng.choice(DELTAS)
            r = rng.random()
            if r < 0.45:
                k = rng.randint(1, 6)
                d = delta
                for _ in range(k):
                    both(f"update_all({d:.5f})", lambda g, p: g.update_all(d, p))
            elif r < 0.70:
                reps = rng.randint(1, 4)
                d = delta
                for _ in range(reps):
                    drain = count if rng.random() < 0.8 else rng.randint(0, count)
                    def reference(g, p):
                        g.update(d, p)
                        for _ in range(drain):
                            g._process(0.0)
                    both(f"update({d:.5f}) + {drain} x process", reference)
            elif r < 0.78:
                frames = rng.randint(1, 7)
                if rng.random() < 0.12:   # a run that crosses a 64-tick block of the cascade-major stream / several deep tick groups
                    frames = rng.randint(60, 70)
                d = delta
                both(f"run({frames})", lambda g, p: g.run(d, p, frames))
                if rng.random() < 0.5:   # runs that follow runs: the last launch of one works ahead for the next, which resumes in the middle of the stream
                    for _ in range(rng.randint(1, 3)):
                        more = rng.choice((frames, frames, rng.randint(1, 9), rng.randint(10, 30)))
                        both(f"run({more}) after a run", lambda g, p: g.run(d, p, more))
                        if rng.random() < 0.2:   # reading the maps in between disturbs nothing
                            compare(a, b, count, f"between runs at step {step}")
            elif r < 0.84:
                i, t = rng.randrange(count), (rng.uniform(8.0, 300.0), rng.uniform(8.0, 300.0))
                def edit(g, p):
                    p[i].tile_length = t
                both(f"tile_length[{i}]", edit)
            elif r < 0.89:
                i, w = rng.randrange(count), rng.uniform(0.0, 2.0)
                def edit(g, p):
                    p[i]._whitecap = w
                both(f"whitecap[{i}]", edit)
            elif r < 0.93 and count > 1:
                fewer, k, d = rng.randint(1, count - 1), rng.randint(1, 4), delta
                for _ in range(k):
                    both(f"update_all of {fewer}", lambda g, p: g.update_all(d, p[:fewer]))
            elif r < 0.96:
                i = rng.randrange(count)
                a.sync()
                saved = a.get_maps(i)[1].copy()
                both(f"set_normal_map[{i}]", lambda g, p: g.set_normal_map(i, saved))
            else:
                compare(a, b, count, f"at step {step}")
        # whatever is still armed is flushed by one more update; then everything must agree
        both("closing update_all", lambda g, p: g.update_all(UPDATE_DELTA, p))
        compare(a, b, count, "at the end")
        assert [p.time for p in pa] == [p.time for p in pb]
    except Exception:
        print(f"FAILED n={n} count={count} seed={seed}; last calls: {log[-12:]}", flush=True)
        raise
    hits = a.lookahead_stats()[0]
    a.free(); b.free()
    return len(log), hits


if __name__ == "__main__":
    if "--caller-stream" in sys.argv:
        sys.argv.remove("--caller-stream")
        CALLER_STREAM = True
    per_config = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    total = served = 0
    for n, count in CONFIGS:
        calls = hits = 0
        for s in range(per_config):
            c, h = schedule(n, count, seed0 + s, ops=40 if n >= 2048 else 60)
            calls += c; hits += h
        print(f"{n}^2 x {count}: {per_config} schedules, {calls} calls, {hits} served from work computed ahead: bit-identical", flush=True)
        total += calls; served += hits
    print(f"all {len(CONFIGS) * per_config} schedules bit-identical ({total} calls, {served} served from work computed ahead)")
