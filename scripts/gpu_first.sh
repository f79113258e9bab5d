#!/bin/bash
# one GPU-box visit: parity tests, smoke, bench (+ CPU baseline), kernel-trace profile of the same bench command
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
rocminfo | grep -E "Marketing|gfx" | head -4 > gpurun_out/gpu.txt 2>&1
nproc >> gpurun_out/gpu.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1
rm -rf gpurun_out/prof1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof1" -o r01 -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline) > gpurun_out/prof.log 2>&1
python scripts/rocprof_summary.py gpurun_out/prof1 gpurun_out/prof1_summary.txt
tail -3 gpurun_out/pytest.log; tail -2 gpurun_out/smoke.log; tail -1 gpurun_out/bench.log; grep -E "^\{" gpurun_out/prof.log | tail -1 | cut -c1-400; head -6 gpurun_out/prof1_summary.txt | cut -c1-140
