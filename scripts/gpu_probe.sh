#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 800 python -m pytest tests -m gpu -q 2>&1 | tail -3
timeout 300 python scripts/mode_bench.py 256:1 128:4 128:1 2>&1 | grep -E "None"
