#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 200 ./tools/kbench 4 30 > gpurun_out/kbench.log 2>&1; grep -A12 "k_pass1c phases" gpurun_out/kbench.log
