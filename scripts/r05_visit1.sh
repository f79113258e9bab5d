#!/bin/bash
# round 5, visit 1: sensors probe, instruction-identity checks, the GPU suite, same-lease A/B of the rounds' libraries and of HEAD's variants, the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_v1; mkdir -p $O; export TMPDIR=/tmp
python - > $O/sensors.txt 2>&1 <<'PY'
import time, json
import amdsmi
amdsmi.amdsmi_init()
hs = amdsmi.amdsmi_get_processor_handles()
print("handles", len(hs))
t0 = time.perf_counter(); m = amdsmi.amdsmi_get_gpu_metrics_info(hs[0]); dt = time.perf_counter() - t0
print("gpu_metrics call ms", dt * 1e3)
print(json.dumps({k: (v if not isinstance(v, (list, tuple)) else list(v)[:10]) for k, v in m.items()}, default=str)[:6000])
PY
ls /sys/class/drm/ >> $O/sensors.txt 2>&1
timeout 300 tools/cvtcheck > $O/cvtcheck.txt 2>&1; echo "cvtcheck rc=$?" >> $O/cvtcheck.txt; cat $O/cvtcheck.txt
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
V=godotoceanwaves_amd/csrc/build/variants
timeout 900 python scripts/ab_rounds.py --cycles 3 r02=tools/ab_rounds/r02 r03=tools/ab_rounds/r03 r04=tools/ab_rounds/r04 base=.:$V/base.so fat=.:$V/fat.so nocmul=.:$V/nocmul.so nocvt=.:$V/nocvt.so head=. > $O/ab_rounds_1024x4.txt 2>&1; tail -9 $O/ab_rounds_1024x4.txt
timeout 600 python scripts/ab_rounds.py --cycles 3 --unmerged r02=tools/ab_rounds/r02 r04=tools/ab_rounds/r04 base=.:$V/base.so head=. > $O/ab_rounds_1024x4_unmerged.txt 2>&1; tail -5 $O/ab_rounds_1024x4_unmerged.txt
timeout 600 python scripts/ab_rounds.py --cycles 2 --config 2048:4 --ticks 300 --reps 5 r04=tools/ab_rounds/r04 base=.:$V/base.so head=. > $O/ab_rounds_2048x4.txt 2>&1; tail -4 $O/ab_rounds_2048x4.txt
timeout 600 python scripts/ab_rounds.py --cycles 2 --config 256:4 --ticks 20000 --reps 5 r04=tools/ab_rounds/r04 base=.:$V/base.so head=. > $O/ab_rounds_256x4.txt 2>&1; tail -4 $O/ab_rounds_256x4.txt
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; tail -c 1500 $O/bench_driver_cmd.json; tail -5 $O/bench_driver_cmd.err
