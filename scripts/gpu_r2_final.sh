#!/bin/bash
# round-2 closing visit: full GPU suite, smoke, default bench line, 2-rank rehearsals of the gather path (with tick groups under it)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest.log; tail -4 gpurun_out/pytest.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_driver_like.log 2>&1; tail -1 gpurun_out/bench_driver_like.log | cut -c1-250
for mode in all root; do
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --share-gpu \
     --cascades 1 --gather-every 8 --gather $mode --steps 96 --warmup 10 --min-time 0.1 --no-cpu-baseline > gpurun_out/rehearsal_$mode.log 2>&1; echo "rehearsal $mode exit $?"; grep '^{' gpurun_out/rehearsal_$mode.log | cut -c1-300
done
