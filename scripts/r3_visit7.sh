timeout 900 python -m pytest tests/test_tick_groups.py tests/test_bench_contract.py tests/test_multi_gpu_host.py -m gpu -x -q 2>&1 | tail -4
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3v7/bench_driver_cmd.json 2> gpurun_out/r3v7/bench.err; echo "bench rc=$?"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --backend gloo --share-gpu > gpurun_out/r3v7/rehearsal2.json 2> gpurun_out/r3v7/rehearsal2.err; echo "rehearsal rc=$?"
python - <<'PY'
import json
for f in ('gpurun_out/r3v7/bench_driver_cmd.json','gpurun_out/r3v7/rehearsal2.json'):
    d=json.loads([l for l in open(f) if l.startswith('{')][0])
    print(f, d['metric'][:90], d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('unmerged',{}).get('value'), d.get('gather'), d.get('no_gather'), d.get('gather_every_tick'))
PY
tail -3 gpurun_out/r3v7/rehearsal2.err
