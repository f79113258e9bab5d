#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
V=$GRAFT_REPO_ROOT/godotoceanwaves_amd/csrc/build/variants/oldorder.so
for rep in 1 2; do
echo "== new order (C0, C2, C1)"; timeout 300 python scripts/mode_bench.py 1024:4 1024:2 2048:1 2>&1 | grep -E "None"
echo "== old order (C0, C1, C2)"; OCEAN_WAVES_LIB=$V timeout 300 python scripts/mode_bench.py 1024:4 1024:2 2048:1 2>&1 | grep -E "None"
done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "compact or batched or runtime" 2>&1 | tail -3
