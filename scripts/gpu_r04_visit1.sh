#!/bin/bash
# round 4, first visit: the whole GPU suite (ABI 4, new parity / hand-off / fault tests, 2048^2 tick pairs), the 2048^2 microbenchmark
# with phase stamps, the lone-tick launch-shape A/B, the 2048^2 configs through bench.py (merged and one launch per pass on one line), a
# slice of the un-rounded FP64 fuzz at 1024 / 2048
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_v1; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log
timeout 300 tools/kbench_2048pair 4 40 > $O/kbench_2048pair_x4.txt 2>&1; cat $O/kbench_2048pair_x4.txt
timeout 200 tools/kbench_2048pair 1 40 > $O/kbench_2048pair_x1.txt 2>&1; head -12 $O/kbench_2048pair_x1.txt
timeout 600 python scripts/lone_tick.py > $O/lone_tick.txt 2>&1; cat $O/lone_tick.txt
for c in 1 2 4 8; do
  timeout 300 python bench.py --map-size 2048 --cascades $c --no-cpu-baseline --steps 300 --warmup 30 > $O/bench_2048x$c.json 2> $O/bench_2048x$c.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_2048x$c.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("2048 x $c", d["ms_per_step"], r["kernel"], r["frac"], "tick", r["tick"]["frac"], "unmerged", r.get("unmerged",{}).get("ms_per_step"), r.get("unmerged",{}).get("frac"))
except Exception as e:
    print("2048 x $c failed", e, open("$O/bench_2048x$c.err").read()[-800:])
PY
done
timeout 600 python scripts/fuzz_parity.py 6 11 --big > $O/fuzz_big.txt 2>&1; tail -8 $O/fuzz_big.txt
