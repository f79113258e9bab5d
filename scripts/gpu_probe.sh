#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 ./tools/kbench 4 200 0 2>&1 | grep -E "two-stream|var  0|^tick  " > gpurun_out/kbench_2s.log
cat gpurun_out/kbench_2s.log
