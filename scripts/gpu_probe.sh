#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 300 --warmup 20 --backend gloo --share-gpu 2>&1 | grep -E '^\{' | cut -c1-400
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 300 --warmup 20 --backend gloo --share-gpu --cascades 1 --gather-every 50 2>&1 | grep -E '^\{' | cut -c1-400
