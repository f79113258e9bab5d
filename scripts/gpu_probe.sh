#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
for m in 0 1 2 3; do timeout 300 ./tools/kbench 4 100 $m 2>&1 | grep -E "data mode|var  0|var  3|var  7|tick"; done > gpurun_out/kbench_modes.log
cat gpurun_out/kbench_modes.log
