#!/bin/bash
# round-2 first visit: GPU tests (new: golden 1024 on the compact kernels, runtime contract, RCCL world-1 gather, C++ host),
# small-config and 2048 phase stamps, a short bench line
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest.log
timeout 200 ./tools/kbench_small_256 4 > gpurun_out/ks_256x4.log 2>&1
timeout 200 ./tools/kbench_small_256 1 > gpurun_out/ks_256x1.log 2>&1
timeout 200 ./tools/kbench_small_512 4 > gpurun_out/ks_512x4.log 2>&1
timeout 300 ./tools/kbench_2048 1 20 > gpurun_out/kbench_2048.log 2>&1
timeout 600 python bench.py --steps 200 --warmup 20 --cpu-seconds 5 > gpurun_out/bench.log 2>&1
tail -15 gpurun_out/pytest.log; cat gpurun_out/ks_256x4.log; cat gpurun_out/ks_256x1.log; tail -40 gpurun_out/kbench_2048.log; tail -2 gpurun_out/bench.log | cut -c1-1500
