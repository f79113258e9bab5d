#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 800 python -m pytest tests -m gpu -q 2>&1 | tail -4
timeout 300 python scripts/mode_bench.py 1024:8 1024:7 1024:6 1024:5 1024:4 2>&1 | grep -E "None"
timeout 200 python bench.py --no-cpu-baseline --cascades 8 2>&1 | tail -1 | cut -c1-200
