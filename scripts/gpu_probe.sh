#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
for rep in 1 2; do
echo "== default mapping"; timeout 300 python scripts/mode_bench.py 1024:4 2048:1 1024:8 2>&1 | grep -E "None"
echo "== map1"; OCEAN_WAVES_LIB=$GRAFT_REPO_ROOT/godotoceanwaves_amd/csrc/build/variants/map1.so timeout 300 python scripts/mode_bench.py 1024:4 2048:1 1024:8 2>&1 | grep -E "None"
done
OCEAN_WAVES_LIB=$GRAFT_REPO_ROOT/godotoceanwaves_amd/csrc/build/variants/map1.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "compact" 2>&1 | tail -2
