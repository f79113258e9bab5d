#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 800 python -m pytest tests -m gpu -q 2>&1 | tail -4
timeout 300 python scripts/mode_bench.py 2>&1 | tee gpurun_out/mode_bench.log | grep -E "None"
