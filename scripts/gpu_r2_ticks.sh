#!/bin/bash
# round-2 visit: persistent tick loop -- microbenchmark first (bounded), then its parity tests, the full GPU suite, small-config bench lines
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 120 ./tools/kbench_small_256 4 400 > gpurun_out/ks2_256x4.log 2>&1
timeout 120 ./tools/kbench_small_256 8 400 > gpurun_out/ks2_256x8.log 2>&1
timeout 120 ./tools/kbench_small_512 4 400 > gpurun_out/ks2_512x4.log 2>&1
head -12 gpurun_out/ks2_256x4.log; grep -E "tick loop|status|^tick" gpurun_out/ks2_256x8.log gpurun_out/ks2_512x4.log
timeout 600 python -m pytest tests/test_tick_loop.py -x -q > gpurun_out/pytest_ticks.log 2>&1; echo "exit $?" >> gpurun_out/pytest_ticks.log; tail -15 gpurun_out/pytest_ticks.log
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest.log; tail -6 gpurun_out/pytest.log
for cfg in "256 4" "256 1" "512 4" "1024 1"; do set -- $cfg
  timeout 300 python bench.py --map-size $1 --cascades $2 --steps 1000 --warmup 50 --no-cpu-baseline > gpurun_out/bench_$1x$2.log 2>&1; tail -1 gpurun_out/bench_$1x$2.log | cut -c1-330
done
