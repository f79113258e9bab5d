#!/bin/bash
# round 5, visit 2: bit-identity of the builds, pairwise pass-1 arithmetic A/B, ow_run's cost per call, the 2048^2 per-cascade look-ahead probe,
# the GPU suite, parity margins, the bench line, a kernel trace of driver-sized regions
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_v2; mkdir -p $O; export TMPDIR=/tmp
V=godotoceanwaves_amd/csrc/build/variants
timeout 900 python scripts/hash_maps.py r04=tools/ab_rounds/r04 base=.:$V/base.so nopw=.:$V/nopw.so head=. > $O/hash_maps.txt 2>&1; tail -12 $O/hash_maps.txt
timeout 600 python scripts/ab_rounds.py --cycles 3 base=.:$V/base.so nopw=.:$V/nopw.so head=. > $O/ab_1024x4.txt 2>&1; tail -4 $O/ab_1024x4.txt
timeout 600 python scripts/ab_rounds.py --cycles 3 --unmerged base=.:$V/base.so nopw=.:$V/nopw.so head=. > $O/ab_1024x4_unmerged.txt 2>&1; tail -4 $O/ab_1024x4_unmerged.txt
timeout 600 python scripts/ab_rounds.py --cycles 2 --config 2048:4 --ticks 300 --reps 5 base=.:$V/base.so nopw=.:$V/nopw.so head=. > $O/ab_2048x4.txt 2>&1; tail -4 $O/ab_2048x4.txt
timeout 600 python scripts/ab_rounds.py --cycles 2 --config 512:8 --ticks 3000 --reps 5 base=.:$V/base.so nopw=.:$V/nopw.so head=. > $O/ab_512x8.txt 2>&1; tail -4 $O/ab_512x8.txt
timeout 600 python scripts/run_overhead.py > $O/run_overhead.txt 2>&1; cat $O/run_overhead.txt
timeout 600 python scripts/la2048_probe.py > $O/la2048_probe.txt 2>&1; cat $O/la2048_probe.txt
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 900 python scripts/parity_margins.py > $O/parity_margins.txt 2>&1; grep "^==" $O/parity_margins.txt
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; tail -c 600 $O/bench_driver_cmd.json; tail -4 $O/bench_driver_cmd.err
for cfg in "1024 8 run" "1024 4 run" "1024 4 unmerged"; do set -- $cfg
  rm -rf /tmp/tr; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python "$GRAFT_REPO_ROOT/scripts/drive_regions.py" --map-size $1 --cascades $2 --mode $3 --ticks 20) > $O/trace_$1x$2_$3.log 2>&1
  python scripts/trace_regions.py /tmp/tr >> $O/trace_$1x$2_$3.log 2>&1; tail -12 $O/trace_$1x$2_$3.log
done
