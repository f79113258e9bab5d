#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
for v in product p2pre product p2pre; do
  if [ $v = product ]; then unset OCEAN_WAVES_LIB; else export OCEAN_WAVES_LIB=$PWD/godotoceanwaves_amd/csrc/build/variants/$v.so; fi
  python scripts/split_variants.py 1024:4 1024:8 512:8 2048:1 2048:4
done 2>&1 | tee gpurun_out/p2c_prefetch.log
export OCEAN_WAVES_LIB=$PWD/godotoceanwaves_amd/csrc/build/variants/p2pre.so
timeout 900 python -m pytest tests -m gpu -q -x -k "compact or golden or many_cascades or full_size or long_run" 2>&1 | tail -3
