#!/bin/bash
# round 4, second visit: 2048^2 tick pairs v2 (8-wave blocks of both passes, half twiddle table) -- microbenchmark with stamps, the 2048^2
# configs through bench.py (pairs | one launch per pass on one line) -- then the whole GPU suite and the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_v2; mkdir -p $O; export TMPDIR=/tmp
timeout 300 tools/kbench_2048pair 4 40 > $O/kbench_2048pair_x4.txt 2>&1; cat $O/kbench_2048pair_x4.txt
timeout 200 tools/kbench_2048pair 1 40 > $O/kbench_2048pair_x1.txt 2>&1; head -8 $O/kbench_2048pair_x1.txt
for c in 1 2 4 8; do
  timeout 300 python bench.py --map-size 2048 --cascades $c --no-cpu-baseline --steps 300 --warmup 30 --min-time 0.5 > $O/bench_2048x$c.json 2> $O/bench_2048x$c.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_2048x$c.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("2048 x $c", d["ms_per_step"], r["kernel"], r["frac"], "tick", r["tick"]["frac"], "unmerged", r.get("unmerged",{}).get("ms_per_step"), r.get("unmerged",{}).get("frac"), r.get("unmerged",{}).get("kernels"))
except Exception as e:
    print("2048 x $c failed", e, open("$O/bench_2048x$c.err").read()[-800:])
PY
done
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -40 $O/pytest.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json; tail -3 $O/bench_default.err
