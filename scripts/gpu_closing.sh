#!/bin/bash
# closing state of a round: the GPU suite, the default bench line, the rocprofv3 kernel trace of the same command on the same box (and of the
# 2048^2 x 4 configuration), PMC passes (separate runs, no tracing domains beside them), the sweep over north_star's grid, a slice of the fuzz.
#   R=r06 bash scripts/gpu_closing.sh          PARTS="bench trace" R=r06 bash scripts/gpu_closing.sh   (a subset: tests bench trace trace2048 trace256 pmc sweep fuzz margins overhead scene rehearsal hosts)
cd "$GRAFT_REPO_ROOT" || exit 1
R=${R:-r06}; O=gpurun_out/${R}_closing; mkdir -p $O; export TMPDIR=/tmp
PARTS=${PARTS:-tests bench trace trace2048 trace256 pmc fuzz margins overhead scene rehearsal hosts}   # (sweep: the whole grid rides in the bench line since round 6)
want() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
if want tests; then
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
fi
if want bench; then
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.json; echo
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; tail -c 300 $O/bench_driver_cmd.json; echo
fi
rm -rf $O/trace $O/trace2048 $O/trace256 $O/pmc
if want trace; then
# (the driver's command, without the CPU leg and the nested PMC passes: what BENCH_rNN's kernels looked like on this box)
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/trace" -o t -- python "$GRAFT_REPO_ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-measure-traffic > "$GRAFT_REPO_ROOT/$O/bench_trace_box.json") > $O/trace.log 2>&1
python scripts/rocprof_summary.py $O/trace $O/kernel_trace_1024x4.txt; head -12 $O/kernel_trace_1024x4.txt | cut -c1-150
fi
if want trace2048; then
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/trace2048" -o t -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-measure-traffic --map-size 2048 --steps 200 --warmup 30 --min-time 0.5 --no-scene) > $O/trace2048.log 2>&1
python scripts/rocprof_summary.py $O/trace2048 $O/kernel_trace_2048x4.txt; head -8 $O/kernel_trace_2048x4.txt | cut -c1-150
fi
if want trace256; then
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/trace256" -o t -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-measure-traffic --map-size 256 --steps 1000 --warmup 100 --min-time 0.5 --no-scene) > $O/trace256.log 2>&1
python scripts/rocprof_summary.py $O/trace256 $O/kernel_trace_256x4.txt; head -8 $O/kernel_trace_256x4.txt | cut -c1-150
fi
if want pmc; then
# PMC: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (they do not fit one), --kernel-trace beside them and nothing else
mkdir -p $O/pmc
for cfg in "1024 4 21" "2048 4 65" "2048 1 21" "1024 8 21" "256 4 81"; do
  set -- $cfg
  for ctr in FETCH_SIZE WRITE_SIZE; do
    d=$O/pmc/n$1x$2_$ctr
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d "$GRAFT_REPO_ROOT/$d" -o p -- python "$GRAFT_REPO_ROOT/scripts/drive.py" --map-size $1 --cascades $2 --frames $3 --warmup 2) > $d.log 2>&1
    echo "$cfg $ctr rc=$?"
  done
done
python scripts/rocprof_summary.py $O/pmc $O/pmc_fetch_write.txt
grep -E "^## |FETCH_SIZE|WRITE_SIZE" $O/pmc_fetch_write.txt | grep -E "^## |k_tick|k_pass" | cut -c1-170
fi
rm -rf $O/trace $O/trace2048 $O/trace256 $O/pmc/*/*.db 2>/dev/null; find $O -name "*.db" -delete
if want sweep; then
rm -f $O/sweep_grid.jsonl
timeout 1800 python bench.py --sweep-grid --steps 500 --warmup 50 --min-time 0.3 --secondary-time 0.3 --cpu-seconds 3 --prime-ms 300 --sweep-out $O/sweep_grid.jsonl > $O/sweep.log 2>&1; python - <<PY
import json
for l in open("$O/sweep_grid.jsonl"):
    d=json.loads(l); r=d["roofline"]; print(d["config"]["map_size"], d["config"]["cascades_per_gpu"], d["value"], d["ms_per_step"], r["kernel"], r["frac"], r["tick"]["frac"], r.get("unmerged",{}).get("ms_per_step"), r.get("unmerged",{}).get("frac"), d.get("cpu_baseline",{}).get("value"))
PY
fi
if want fuzz; then
timeout 900 python scripts/fuzz_parity.py 30 501 > $O/fuzz.txt 2>&1; tail -3 $O/fuzz.txt
timeout 900 python scripts/fuzz_schedule.py 12 501 > $O/fuzz_schedule.txt 2>&1; tail -3 $O/fuzz_schedule.txt
timeout 900 python scripts/fuzz_schedule.py 6 321 --caller-stream > $O/fuzz_schedule_caller_stream.txt 2>&1; tail -1 $O/fuzz_schedule_caller_stream.txt
fi
if want margins; then
timeout 900 python scripts/parity_margins.py > $O/parity_margins.txt 2>&1; grep "^==" $O/parity_margins.txt
fi
if want overhead; then
timeout 900 python scripts/run_overhead.py 1024:8 2048:4 256:4 1024:4 > $O/run_overhead.txt 2>&1; grep "ow_run " $O/run_overhead.txt | cut -c1-120
fi
if want scene; then
rm -rf $O/scene_trace
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/scene_trace" -o t -- python "$GRAFT_REPO_ROOT/scripts/scene_probe.py" --hz 144) > $O/scene_probe_144.log 2>&1
python scripts/rocprof_summary.py $O/scene_trace $O/scene_kernel_trace_144hz.txt; grep "per update" $O/scene_probe_144.log; rm -rf $O/scene_trace
fi
if want rehearsal; then
# the N > 1 rank code on the one GPU of the box (gloo, both ranks on GPU 0: control flow only, the numbers mean nothing)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --backend gloo --share-gpu > $O/rehearsal_2rank.json 2> $O/rehearsal_2rank.err; tail -c 600 $O/rehearsal_2rank.json
fi
if want hosts; then
# the compiled C99 host of the device group: two and four shards on the one device, every shard through the peer path; prints the model next to the measurement
gcc -O2 -std=c99 -Iinclude examples/multi_gpu_host.c -o /tmp/multi_gpu_host -Lgodotoceanwaves_amd -locean_waves -Wl,-rpath,$PWD/godotoceanwaves_amd -Wl,-rpath-link,/opt/rocm/lib -lm
{ /tmp/multi_gpu_host 1024 1 400 16 0,0 peer; /tmp/multi_gpu_host 1024 1 400 16 0,0,0,0,0,0,0,0 peer; } > $O/multi_gpu_host.txt 2>&1; cat $O/multi_gpu_host.txt
fi
