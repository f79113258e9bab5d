#!/bin/bash
# round 4, visit 5: the shipped stream order (cascade-major for every multi-batch tick), the live traffic measurement of bench.py, the tick tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_v5; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_tick_groups.py tests/test_bench_contract.py tests/test_gpu_parity.py -m gpu -q --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1]); r=d["roofline"]
print(d["value"], d["ms_per_step"], d["gpu_phase_s"], r["frac"], r["traffic"], r.get("traffic_bytes_per_texel"), r["traffic_source"][:60], r.get("traffic_measurement_failed"))
PY
for cfg in "1024 8" "1024 5" "1024 7" "2048 4"; do set -- $cfg
  timeout 300 python bench.py --map-size $1 --cascades $2 --no-cpu-baseline --steps 300 --warmup 30 --min-time 0.5 > $O/bench_$1x$2.json 2> $O/bench_$1x$2.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$1x$2.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("$1 x $2", d["ms_per_step"], r["kernel"], r["frac"], "unmerged", r.get("unmerged",{}).get("ms_per_step"), r.get("unmerged",{}).get("frac"), "traffic B/texel", r.get("traffic_bytes_per_texel"), r.get("traffic_measurement_failed"))
except Exception as e:
    print("$1 x $2 failed", e, open("$O/bench_$1x$2.err").read()[-600:])
PY
done
