#!/bin/bash
# round-2 visit 10 (closing state: tick pairs incl. 2048^2, residency budget 248 MiB): full GPU suite, smoke, default bench, sweep, kernel trace of 2048^2 x 4
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest.log; tail -3 gpurun_out/pytest.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-300
rm -f gpurun_out/sweep.jsonl
timeout 900 python bench.py --sweep --sweep-out gpurun_out/sweep.jsonl --steps 1000 --warmup 100 > gpurun_out/sweep.log 2>&1; echo "sweep exit $?"
python - <<'PY'
import json
for l in open('gpurun_out/sweep.jsonl'):
    d=json.loads(l); r=d['roofline']
    print(d['config']['map_size'], d['config']['cascades_per_gpu'], 'maps/s', d['value'], 'ms/tick', d['ms_per_step'], r['kernel'], 'avg_launch_ms', r['avg_launch_ms'], 'frac', r['frac'], 'copy', r['frac_of_copy_ceiling'], 'tick frac', r['tick']['frac'], 'copy', r['tick']['frac_of_copy_ceiling'], 'p1', r['pass1_ms'], 'p2', r['pass2_ms'], 'cpu', d['cpu_baseline']['value'])
PY
rm -rf gpurun_out/prof_r02e_2048x4
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r02e_2048x4" -o t -- python "$GRAFT_REPO_ROOT/scripts/drive.py" --map-size 2048 --cascades 4 --frames 400 --warmup 100) > gpurun_out/prof_r02e_2048x4.log 2>&1
python scripts/rocprof_summary.py gpurun_out/prof_r02e_2048x4 gpurun_out/prof_r02e_2048x4_summary.txt; head -6 gpurun_out/prof_r02e_2048x4_summary.txt | cut -c1-150
