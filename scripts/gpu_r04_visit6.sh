#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_v6; mkdir -p $O; export TMPDIR=/tmp
echo "== 2048^2 one launch per pass: k_pass2c 16-wave blocks | pass 2 in 8-wave blocks of 4 columns (OW_DEBUG_P2_QUARTER)" | tee $O/p2_quarter.txt
for c in 1 4; do
  python scripts/pairs_2048.py 2048:$c 2>&1 | head -2 | tee -a $O/p2_quarter.txt
  OW_DEBUG_P2_QUARTER=1 python scripts/pairs_2048.py 2048:$c 2>&1 | head -2 | tee -a $O/p2_quarter.txt
done
echo "== 1024^2 x 4 headline: batches of 4 | 2 | 1 cascades (cascade-major pairs)" | tee $O/headline_batches.txt
for t in 4 2 1; do OW_DEBUG_PAIR_TEXELS=$t python scripts/pairs_2048.py 1024:4 2>&1 | sed -n 1,4p | tee -a $O/headline_batches.txt; done
