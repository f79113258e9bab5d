#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k compact 2>&1 | tail -25
timeout 200 python scripts/mode_bench.py 1024:4 1024:2 2048:1 2>&1 | tail -12
