#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 ./tools/kbench 4 200 0 2>&1 | grep -E "stagger|pass1 var  0|aux T=0 H=0|^tick  " > gpurun_out/kbench_stagger.log; cat gpurun_out/kbench_stagger.log
