#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out; rm -f gpurun_out/bench_sizes.log
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -4
for cfg in "1024 4"; do set -- $cfg; timeout 300 python bench.py --map-size $1 --cascades $2 --steps 1000 --warmup 100 --prime-ms 150 --no-cpu-baseline >> gpurun_out/bench_sizes.log 2>&1; done
python3 - <<'PY'
import json
for l in open('gpurun_out/bench_sizes.log'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['config']['map_size'], d['config']['cascades_per_gpu'], 'maps/s', d['value'], 'us/step', round(d['ms_per_step']*1e3,1), 'p1', r['pass1_ms'], 'p2', r['pass2_ms'], 'tick_frac', r['tick_frac'])
    elif 'rror' in l: print(l.strip()[:300])
PY
