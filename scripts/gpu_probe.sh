#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
V=$GRAFT_REPO_ROOT/godotoceanwaves_amd/csrc/build/variants
echo "== no delay"; timeout 300 python scripts/mode_bench.py 1024:4 2048:1 2>&1 | grep -E "None"
for d in 2 4 6 8; do echo "== lower blocks delayed by ~$d us"; OCEAN_WAVES_LIB=$V/delay$d.so timeout 300 python scripts/mode_bench.py 1024:4 2048:1 2>&1 | grep -E "None"; done
echo "== no delay"; timeout 300 python scripts/mode_bench.py 1024:4 2048:1 2>&1 | grep -E "None"
