#!/bin/bash
# round 6, last session: the wave-pair rendezvous of the 2048^2 rows in two halves (arrive where the reads were issued, wait before the next write; no s_waitcnt in front of the epoch word):
# variant libraries against the in-tree build, hashes first, then us per tick.   B="base=. name=.:lib ..." bash scripts/gpu_ab_rowsync_r06.sh
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_ab_rowsync; rm -rf $O; mkdir -p $O
V=godotoceanwaves_amd/csrc/build/variants
B=${B:-"base=. rsplit=.:$V/rsplit.so rsplitw=.:$V/rsplitw.so"}
timeout 600 python scripts/hash_maps.py --configs 2048:1,2048:2,1024:4 $B > $O/hash.txt 2>&1; echo "hash rc=$?"; tail -5 $O/hash.txt
timeout 900 python scripts/ab_rounds.py --cycles 3 --config 2048:4 --ticks 300 --reps 5 $B > $O/ab_2048x4.txt 2>&1; tail -4 $O/ab_2048x4.txt
timeout 900 python scripts/ab_rounds.py --cycles 3 --config 2048:1 --ticks 1200 --reps 5 $B > $O/ab_2048x1.txt 2>&1; tail -4 $O/ab_2048x1.txt
timeout 900 python scripts/ab_rounds.py --cycles 2 --config 2048:4 --ticks 300 --reps 5 --unmerged $B > $O/ab_2048x4_unmerged.txt 2>&1; tail -4 $O/ab_2048x4_unmerged.txt
