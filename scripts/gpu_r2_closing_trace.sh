#!/bin/bash
# closing state of round 2: default bench line and the rocprofv3 kernel trace of the same command on the same box
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/bench_closing.log 2>&1; tail -1 gpurun_out/bench_closing.log | cut -c1-200
rm -rf gpurun_out/prof_r02f
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r02f" -o r02 -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline) > gpurun_out/prof_r02f.log 2>&1
python scripts/rocprof_summary.py gpurun_out/prof_r02f gpurun_out/prof_r02f_summary.txt; head -8 gpurun_out/prof_r02f_summary.txt | cut -c1-160
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_closing.log').read().strip().splitlines()[-1]); r=d['roofline']
print('bench:', d['value'], 'maps/s', d['ms_per_step'], 'ms/tick;', r['kernel'], 'avg_launch_ms', r['avg_launch_ms'], 'frac', r['frac'], 'traffic', r['traffic'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
