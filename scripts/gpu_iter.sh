#!/bin/bash
# quick iteration visit: ablation microbench, GPU parity tests, short bench
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 ./tools/kbench 4 > gpurun_out/kbench.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest.log
timeout 600 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/bench.log 2>&1
cat gpurun_out/kbench.log; tail -4 gpurun_out/pytest.log; python - <<'PY'
import json
for l in open('gpurun_out/bench.log'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('BENCH maps/s',d['value'],'ms/step',d['ms_per_step'],'p1',r['pass1_ms'],'p2',r['pass2_ms'],'tick_frac',r['tick_frac'])
PY
