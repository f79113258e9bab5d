#!/bin/bash
# round 6: do the 32-byte pieces of the 4-row pass-1 blocks at 2048^2 leave the L2 as 64-byte write requests?  (TCC_EA0_WRREQ / _64B, one pass, --kernel-trace beside it and nothing else)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_pmc_wrreq; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for cfg in "2048 1" "1024 4"; do set -- $cfg
for ctr in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WR_UNCACHED_32B_sum"; do
 tag=$(echo $ctr | cut -d' ' -f1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d "$GRAFT_REPO_ROOT/$O/n$1x$2_$tag" -o p -- python "$GRAFT_REPO_ROOT/scripts/drive.py" --map-size $1 --cascades $2 --frames 21 --warmup 2 --single-stream) > $O/n$1x$2_$tag.log 2>&1; echo "$cfg $tag rc=$?"
done; done
python scripts/rocprof_summary.py $O $O/summary.txt; grep -E "^## |k_tick_pair|k_pass1c|k_pass2c" $O/summary.txt | grep -v "^# " | cut -c1-150; find $O -name "*.db" -delete
