#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
for cfg in "256 4" "1024 1" "1024 8" "2048 4"; do set -- $cfg
  rm -rf gpurun_out/prof_$1x$2
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_$1x$2" -o t -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --map-size $1 --cascades $2 --steps 1000 --warmup 100) > gpurun_out/prof_$1x$2.log 2>&1
  python scripts/rocprof_summary.py gpurun_out/prof_$1x$2 gpurun_out/prof_$1x$2_summary.txt
  echo "== $1^2 x $2"; grep -E '^\{' gpurun_out/prof_$1x$2.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'],'maps/s', d['ms_per_step']*1e3,'us/tick', d['roofline']['kernel'], d['roofline']['pass1_ms']*1e3, d['roofline']['pass2_ms']*1e3)"
  head -6 gpurun_out/prof_$1x$2_summary.txt | tail -3 | cut -c1-150
done
