#!/bin/bash
# round-2 visit 7 (tick pairs on the compact family): tick-group / pair tests, merged-vs-unmerged table, default bench line, bench of the mid configs
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_tick_groups.py tests/test_bench_contract.py tests/test_golden.py -m gpu -q -x > gpurun_out/pytest_pairs.log 2>&1; tail -5 gpurun_out/pytest_pairs.log
timeout 300 python scripts/merged_launches.py 2>&1 | tee gpurun_out/merged_launches.txt
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-1800
for c in 2 3; do timeout 300 python bench.py --cascades $c --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400; done
