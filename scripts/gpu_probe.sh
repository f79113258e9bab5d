#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
for m in 0 2 0 2; do timeout 300 ./tools/kbench 4 200 $m 2>&1 | grep -E "data mode|var  0|^tick  "; done > gpurun_out/kbench_modes.log
cat gpurun_out/kbench_modes.log
timeout 120 python scripts/drive.py --frames 400 --warmup 200
