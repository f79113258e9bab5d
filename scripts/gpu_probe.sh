#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python scripts/gpu_err.py 2>&1 | tee gpurun_out/parity_margins.txt | tail -14
