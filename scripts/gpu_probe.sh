#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
for bt in 4194304 2097152 1048576; do for c in 4 8; do echo "batch texels $bt cascades $c"; OW_BATCH_TEXELS=$bt timeout 120 python scripts/drive.py --map-size 1024 --cascades $c --frames 1500 --warmup 1500; done; done 2>&1 | tee gpurun_out/batch.log
