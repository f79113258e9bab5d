#!/bin/bash
# closing state of a round: default bench line, the rocprofv3 kernel trace of the same command on the same box, PMC passes (separate runs),
# the sweep over the BASELINE configurations, the GPU suite's tail.    R=r03 bash scripts/gpu_closing.sh
cd "$GRAFT_REPO_ROOT" || exit 1
R=${R:-r03}; O=gpurun_out/${R}_closing; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json
rm -rf $O/trace $O/trace2048 $O/pmc1 $O/pmc2
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/trace" -o t -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline) > $O/trace.log 2>&1
python scripts/rocprof_summary.py $O/trace $O/kernel_trace_1024x4.txt; head -12 $O/kernel_trace_1024x4.txt | cut -c1-150
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/trace2048" -o t -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --map-size 2048 --steps 300 --warmup 30) > $O/trace2048.log 2>&1
python scripts/rocprof_summary.py $O/trace2048 $O/kernel_trace_2048x4.txt; head -8 $O/kernel_trace_2048x4.txt | cut -c1-150
# PMC: counters in their own runs, no tracing domains beside them (two passes: the counters do not fit one)
(cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE -d "$GRAFT_REPO_ROOT/$O/pmc1" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-unmerged --steps 200 --warmup 20 --min-time 0.05 --prime-ms 50) > $O/pmc1.log 2>&1
python scripts/rocprof_summary.py $O/pmc1 $O/pmc_fetch_write_1024x4.txt; grep -E "k_tick_pair_c|k_pass" $O/pmc_fetch_write_1024x4.txt | cut -c1-170 | head
timeout 1500 python bench.py --sweep --steps 500 --warmup 50 --sweep-out $O/sweep.jsonl > $O/sweep.log 2>&1; python - <<PY
import json
for l in open("$O/sweep.jsonl"):
    d=json.loads(l); r=d["roofline"]; print(d["config"]["map_size"], d["config"]["cascades_per_gpu"], d["value"], d["ms_per_step"], r["kernel"], r["frac"], r["tick"]["frac"], r.get("unmerged",{}).get("ms_per_step"), d.get("cpu_baseline",{}).get("value"))
PY
