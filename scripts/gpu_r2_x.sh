#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
for v in product nobar nofft; do
  if [ $v = product ]; then unset OCEAN_WAVES_LIB; else export OCEAN_WAVES_LIB=$PWD/godotoceanwaves_amd/csrc/build/variants/$v.so; fi
  python scripts/split_variants.py 1024:4 1024:2
done 2>&1 | tee gpurun_out/p1c_variants.log
