#!/bin/bash
# round 6, generic visit: PARTS="new suite bench overhead kbench" OUT=name bash scripts/r06_visit.sh   (logs under gpurun_out/$OUT)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=${OUT:-r06_v1}; O=gpurun_out/$OUT; mkdir -p $O; export TMPDIR=/tmp
PARTS=${PARTS:-new suite bench overhead kbench}
want() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
if want fix; then
timeout 900 python -m pytest tests/test_lookahead.py tests/test_multi_gpu_host.py "tests/test_bench_contract.py::test_one_json_line_with_the_contract_keys" -m gpu -q --timeout 600 > $O/pytest_fix.log 2>&1; echo "pytest rc=$?" >> $O/pytest_fix.log; tail -12 $O/pytest_fix.log
fi
if want new; then   # what this round changed, first and without -x: every failure is worth seeing
timeout 1500 python -m pytest tests/test_run_after_run.py tests/test_spectrum_resident.py tests/test_parity_margins.py tests/test_tick_groups.py \
  "tests/test_group.py::test_link_info_says_how_each_shard_reaches_the_root" "tests/test_gpu_parity.py::test_baseline_configs_through_ow_run_match_the_oracle" \
  "tests/test_lookahead.py" -m gpu -q --timeout 900 > $O/pytest_new.log 2>&1; echo "pytest rc=$?" >> $O/pytest_new.log; tail -30 $O/pytest_new.log
fi
if want suite; then
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log
fi
if want bench; then
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; tail -c 2500 $O/bench_driver_cmd.json; tail -5 $O/bench_driver_cmd.err
fi
if want overhead; then
timeout 900 python scripts/run_overhead.py 1024:8 2048:4 256:4 1024:4 > $O/run_overhead.txt 2>&1; cat $O/run_overhead.txt
fi
if want kbench; then
: > $O/kbench.txt
for b in ${KBENCH:-exp bar}; do
  for cs in ${KBENCH_C:-1 4}; do
    echo "== tools/kbench_2048pair_$b $cs 40" >> $O/kbench.txt
    timeout 300 tools/kbench_2048pair_$b $cs 40 2>&1 | grep -v "clocks\|waves\|issued\|table\|modulated\|input\|transformed\|staged\|acknowledged" >> $O/kbench.txt
  done
done
cat $O/kbench.txt
fi
if want san; then
OUT=$OUT bash scripts/run_sanitized.sh > $O/san_stdout.log 2>&1; tail -70 $O/san_stdout.log
fi
if want tsan; then
OUT=$OUT ONLY_TSAN=1 bash scripts/run_sanitized.sh > $O/tsan_stdout.log 2>&1; tail -40 $O/tsan_stdout.log
fi
if want scene; then   # the scene's cadence under a kernel trace: launches per update, and what each costs
rm -rf $O/scene_trace
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/scene_trace" -o t -- python "$GRAFT_REPO_ROOT/scripts/scene_probe.py" --hz 144) > $O/scene_probe_144.log 2>&1
python scripts/rocprof_summary.py $O/scene_trace $O/scene_kernel_trace_144hz.txt; grep "per update" $O/scene_probe_144.log; head -8 $O/scene_kernel_trace_144hz.txt | cut -c1-160
rm -rf $O/scene_trace
timeout 300 python scripts/scene_probe.py --hz 60 | grep "per update"
fi
if want spectrum; then   # k_spectrum under a kernel trace, and its parity tests
timeout 900 python -m pytest "tests/test_gpu_parity.py" -m gpu -q -k "spectrum or dirty or edges" --timeout 600 > $O/pytest_spectrum.log 2>&1; tail -3 $O/pytest_spectrum.log
rm -rf $O/spec_trace
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/spec_trace" -o t -- python "$GRAFT_REPO_ROOT/scripts/drive.py" --map-size 1024 --cascades 8 --frames 4) > $O/spec_trace.log 2>&1
python scripts/rocprof_summary.py $O/spec_trace $O/spectrum_kernel_trace.txt; grep -i "spectrum" $O/spectrum_kernel_trace.txt | cut -c1-120
rm -rf $O/spec_trace
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/spec_trace" -o t -- python "$GRAFT_REPO_ROOT/scripts/drive.py" --map-size 2048 --cascades 4 --frames 4) > $O/spec_trace.log 2>&1
python scripts/rocprof_summary.py $O/spec_trace $O/spectrum_kernel_trace_2048.txt; grep -i "spectrum" $O/spectrum_kernel_trace_2048.txt | cut -c1-120
rm -rf $O/spec_trace
fi
if want hash; then
timeout 900 python scripts/hash_maps.py ${HASH_BUILDS} > $O/hash_maps.txt 2>&1; cat $O/hash_maps.txt
fi
if want ab; then
for cfg in ${AB_CONFIGS:-2048:4 1024:4}; do
  t=2000; [ "${cfg%%:*}" = "2048" ] && t=300
  timeout 1500 python scripts/ab_rounds.py --cycles ${AB_CYCLES:-3} --config $cfg --ticks $t --reps 5 ${AB_BUILDS} > $O/ab_$cfg.txt 2>&1; tail -12 $O/ab_$cfg.txt
done
fi
if want fuzz; then
timeout 1500 python scripts/fuzz_schedule.py ${FUZZ_N:-12} ${FUZZ_SEED:-601} > $O/fuzz_schedule.txt 2>&1; tail -14 $O/fuzz_schedule.txt
fi
