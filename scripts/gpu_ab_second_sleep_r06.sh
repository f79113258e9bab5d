cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_ab_second_sleep; rm -rf $O; mkdir -p $O
B="base=. ss6k=.:godotoceanwaves_amd/csrc/build/variants/ss6000.so ss12k=.:godotoceanwaves_amd/csrc/build/variants/ss12000.so ss20k=.:godotoceanwaves_amd/csrc/build/variants/ss20000.so"
timeout 600 python scripts/hash_maps.py --configs 2048:1 $B > $O/hash.txt 2>&1; tail -2 $O/hash.txt
timeout 900 python scripts/ab_rounds.py --cycles 3 --config 2048:4 --ticks 300 --reps 5 $B > $O/ab_2048x4.txt 2>&1; tail -5 $O/ab_2048x4.txt
timeout 900 python scripts/ab_rounds.py --cycles 3 --config 2048:1 --ticks 1200 --reps 5 $B > $O/ab_2048x1.txt 2>&1; tail -5 $O/ab_2048x1.txt
