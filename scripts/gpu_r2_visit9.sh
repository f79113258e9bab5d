#!/bin/bash
# round-2 visit 9 (tick pairs shipped): full GPU suite, smoke, default bench, sweep, kernel traces (headline through bench.py, other configs through
# drive.py), PMC passes for k_tick_pair_c at 1024^2 x 4
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest.log; tail -4 gpurun_out/pytest.log
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
rm -rf gpurun_out/prof_r02d
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r02d" -o r02 -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline) > gpurun_out/prof_r02d.log 2>&1
python scripts/rocprof_summary.py gpurun_out/prof_r02d gpurun_out/prof_r02d_summary.txt; head -8 gpurun_out/prof_r02d_summary.txt | cut -c1-160
for cfg in "1024 2" "1024 1" "512 8" "1024 8"; do set -- $cfg
  rm -rf gpurun_out/prof_r02d_$1x$2
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r02d_$1x$2" -o t -- python "$GRAFT_REPO_ROOT/scripts/drive.py" --map-size $1 --cascades $2 --frames 400 --warmup 100) > gpurun_out/prof_r02d_$1x$2.log 2>&1
  python scripts/rocprof_summary.py gpurun_out/prof_r02d_$1x$2 gpurun_out/prof_r02d_$1x$2_summary.txt; echo "== $1 x $2"; head -5 gpurun_out/prof_r02d_$1x$2_summary.txt | cut -c1-150
done
mkdir -p gpurun_out/pmc_r02d
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_r02d/1024x4_p$i" -o p$i -- python "$GRAFT_REPO_ROOT/scripts/drive.py" --map-size 1024 --cascades 4 --frames 21 --warmup 2) > gpurun_out/pmc_r02d/1024x4_p$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
python scripts/rocprof_summary.py gpurun_out/pmc_r02d gpurun_out/pmc_r02d_summary.txt; grep -E "k_tick_pair.*(FETCH_SIZE|WRITE_SIZE|TCC)" gpurun_out/pmc_r02d_summary.txt | cut -c1-200
