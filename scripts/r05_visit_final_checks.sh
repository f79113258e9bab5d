#!/bin/bash
# round 5, last checks of the shipped build: bit-identity to round 4's library, the full 320-schedule fuzz, random FP64 records, a last A/B against round 4's kernels
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_final; mkdir -p $O; export TMPDIR=/tmp
V=godotoceanwaves_amd/csrc/build/variants
timeout 600 python scripts/hash_maps.py r04=tools/ab_rounds/r04 head=. > $O/hash_maps.txt 2>&1; tail -11 $O/hash_maps.txt
timeout 1500 python scripts/fuzz_schedule.py 40 100 > $O/fuzz_schedule.txt 2>&1; tail -10 $O/fuzz_schedule.txt
timeout 900 python scripts/fuzz_parity.py 60 401 > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
timeout 900 python scripts/fuzz_parity.py 40 77 --wilder --small > $O/fuzz_wilder.txt 2>&1; tail -2 $O/fuzz_wilder.txt
timeout 600 python scripts/ab_rounds.py --cycles 2 r04=tools/ab_rounds/r04 head=. > $O/ab_final.txt 2>&1; tail -3 $O/ab_final.txt
