#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "compact or batched" 2>&1 | tail -5
timeout 300 python scripts/mode_bench.py 1024:4 1024:2 2048:1 2>&1 | grep -E "None|standard"
