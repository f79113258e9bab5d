#!/bin/bash
# round-2 visit: split plan for 2048 (k_pass1c_split / k_pass2c_split): parity first, then timings
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "2048 or full_size or many_cascades or rendezvous or foam_state" > gpurun_out/pytest_2048.log 2>&1; echo "exit $?" >> gpurun_out/pytest_2048.log; tail -25 gpurun_out/pytest_2048.log
timeout 300 python scripts/mode_bench.py 2048:1 2048:4 2>&1 | grep -E "None|standard   |compact  " > gpurun_out/mode_2048.log; cat gpurun_out/mode_2048.log
