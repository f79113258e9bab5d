#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python scripts/mode_bench.py 2>&1 | tee gpurun_out/mode_bench.log | tail -40
