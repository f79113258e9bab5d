#!/bin/bash
# round-2 visit 8 (tick pairs over two batches per tick): tests, merged-vs-unmerged table, sweep
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_tick_groups.py tests/test_bench_contract.py -m gpu -q -x > gpurun_out/pytest_pairs.log 2>&1; tail -5 gpurun_out/pytest_pairs.log
timeout 300 python scripts/merged_launches.py 1024:4 1024:5 1024:6 1024:7 1024:8 512:8 2>&1 | tee gpurun_out/merged_launches2.txt
timeout 300 python bench.py --cascades 8 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-2200
