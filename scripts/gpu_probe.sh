#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python scripts/gpu_err.py > gpurun_out/err.log 2>&1; cat gpurun_out/err.log

timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -5
