#!/bin/bash
# round 3, visit 1: full GPU suite (new: test_group, contract changes, golden fixtures of the whole headline batch), default bench, 2-rank rehearsal
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3v1
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3v1/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3v1/pytest.log
tail -15 gpurun_out/r3v1/pytest.log
timeout 300 python bench.py > gpurun_out/r3v1/bench_default.json 2> gpurun_out/r3v1/bench_default.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/r3v1/bench_default.json
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --backend gloo --share-gpu --steps 400 --warmup 20 --min-time 0.2 > gpurun_out/r3v1/rehearsal2.json 2> gpurun_out/r3v1/rehearsal2.err; echo "rehearsal rc=$?"
tail -c 1500 gpurun_out/r3v1/rehearsal2.json; tail -5 gpurun_out/r3v1/rehearsal2.err
