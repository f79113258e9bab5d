#!/bin/bash
# round 6, last session: same-lease A/B of builds that differ in COMPILER SCHEDULING FLAGS only (never tried before) and of LDS slot map 2 again
# (round 3 measured it when the kernels were 10 % slower): scripts/build_variant.sh <name> <flags> beforehand.  Hashes first (bit-identical or not), then us per tick.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_ab_flags; rm -rf $O; mkdir -p $O
V=godotoceanwaves_amd/csrc/build/variants
B="base=. map2=.:$V/map2.so memcl=.:$V/memcl.so nopost=.:$V/nopost.so wprio=.:$V/wprio.so"
timeout 600 python scripts/hash_maps.py --configs 1024:4,2048:1,256:4 $B > $O/hash.txt 2>&1; echo "hash rc=$?"; tail -8 $O/hash.txt
timeout 900 python scripts/ab_rounds.py --cycles 3 --config 2048:4 --ticks 300 --reps 5 $B > $O/ab_2048x4.txt 2>&1; tail -6 $O/ab_2048x4.txt
timeout 900 python scripts/ab_rounds.py --cycles 3 --config 1024:4 --ticks 2000 --reps 5 $B > $O/ab_1024x4.txt 2>&1; tail -6 $O/ab_1024x4.txt
timeout 900 python scripts/ab_rounds.py --cycles 3 --config 256:4 --ticks 12000 --reps 5 $B > $O/ab_256x4.txt 2>&1; tail -6 $O/ab_256x4.txt
