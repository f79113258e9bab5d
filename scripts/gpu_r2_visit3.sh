#!/bin/bash
# round-2 visit 3: kernel arguments fetched in one round trip (stamps), full GPU suite, 2-rank rehearsal of the gather path, config sweep
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 120 ./tools/kbench_small_256 4 1000 > gpurun_out/ks3_256x4.log 2>&1; grep -E "^tick|alone|loads issued|acknowledged" gpurun_out/ks3_256x4.log
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest.log; tail -8 gpurun_out/pytest.log
for mode in all root; do
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --share-gpu \
     --cascades 1 --gather-every 1 --gather $mode --steps 100 --warmup 10 --min-time 0.1 --no-cpu-baseline > gpurun_out/rehearsal_$mode.log 2>&1; echo "rehearsal $mode exit $?"; grep '^{' gpurun_out/rehearsal_$mode.log | cut -c1-600
done
rm -f gpurun_out/sweep.jsonl
timeout 900 python bench.py --sweep --sweep-out gpurun_out/sweep.jsonl --steps 500 --warmup 50 > gpurun_out/sweep.log 2>&1; echo "sweep exit $?"
python - <<'PY'
import json
for l in open('gpurun_out/sweep.jsonl'):
    d=json.loads(l); r=d['roofline']
    print(d['config']['map_size'], d['config']['cascades_per_gpu'], 'maps/s', d['value'], 'ms/tick', d['ms_per_step'], r['kernel'], 'frac', r['frac'], 'tick frac', r['tick']['frac'], 'copy', r['tick']['frac_of_copy_ceiling'], 'p1', r['pass1_ms'], 'p2', r['pass2_ms'], 'cpu', d['cpu_baseline']['value'])
PY
