#!/bin/bash
# round 4, visit 4: the lone-small-tick-in-one-launch experiment (tools/kbench_solo), the batch-order A/B of multi-batch ticks at 1024^2,
# and the PMC passes the closing visit lost to a missing directory
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_v4; mkdir -p $O/pmc; export TMPDIR=/tmp
for c in 4 1 8; do timeout 200 tools/kbench_solo_256 $c 2000 > $O/solo_256x$c.txt 2>&1; cat $O/solo_256x$c.txt; done
for c in 1 4; do timeout 200 tools/kbench_solo_512 $c 1000 > $O/solo_512x$c.txt 2>&1; cat $O/solo_512x$c.txt; done
timeout 600 python scripts/pairs_2048.py 1024:8 1024:6 1024:5 512:8 > $O/pairs_1024_order.txt 2>&1; cat $O/pairs_1024_order.txt
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
find $O -name "*.db" -delete
