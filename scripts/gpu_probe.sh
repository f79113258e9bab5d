#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
V=$GRAFT_REPO_ROOT/godotoceanwaves_amd/csrc/build/variants/hwsin.so
for rep in 1 2; do
echo "== polynomial"; timeout 300 python scripts/mode_bench.py 1024:4 1024:1 256:4 2>&1 | grep -E "None"
echo "== hw sincos"; OCEAN_WAVES_LIB=$V timeout 300 python scripts/mode_bench.py 1024:4 1024:1 256:4 2>&1 | grep -E "None"
done
OCEAN_WAVES_LIB=$V timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -q -x 2>&1 | tail -3
OCEAN_WAVES_LIB=$V python scripts/gpu_err.py 2>&1 | tail -12
