#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
for v in product maxd6 maxd8 maxd12; do
  if [ $v = product ]; then unset OCEAN_WAVES_LIB; else export OCEAN_WAVES_LIB=$PWD/godotoceanwaves_amd/csrc/build/variants/$v.so; fi
  echo "== $v"; timeout 600 python scripts/pairs_crossover.py 256:1 256:4 256:8 512:2 512:4 1024:1 2>&1 | cut -c1-45
done > gpurun_out/groups_depth2.log 2>&1; cat gpurun_out/groups_depth2.log
