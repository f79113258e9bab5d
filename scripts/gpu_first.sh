#!/bin/bash
# one GPU-box visit: parity tests, smoke, bench, kernel-trace profile
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
rocminfo | grep -E "Marketing|gfx" | head -4 > gpurun_out/gpu.txt 2>&1
nproc >> gpurun_out/gpu.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 600 python bench.py --steps 200 --warmup 10 > gpurun_out/bench.log 2>&1
for cfg in "256 4" "512 4" "2048 4" "1024 1" "1024 8"; do set -- $cfg; timeout 300 python bench.py --map-size $1 --cascades $2 --steps 100 --warmup 5 --no-cpu-baseline >> gpurun_out/bench_sizes.log 2>&1; done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof1" -o r01 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 100 --warmup 5 --no-cpu-baseline) > gpurun_out/prof.log 2>&1
tail -3 gpurun_out/pytest.log; cat gpurun_out/smoke.log | tail -2; cat gpurun_out/bench.log | tail -2
