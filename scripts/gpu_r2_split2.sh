#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
python scripts/split_variants.py 2048:2 2048:4 1024:8 > gpurun_out/split_variants.log 2>&1
for v in batch8 batch16; do OCEAN_WAVES_LIB=$PWD/godotoceanwaves_amd/csrc/build/variants/$v.so python scripts/split_variants.py 2048:2 2048:4 1024:8 >> gpurun_out/split_variants.log 2>&1; done
cat gpurun_out/split_variants.log
