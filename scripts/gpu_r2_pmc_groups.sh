#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the tick-group kernel (pipelined pass 2) at 256^2 x 4 and of the plain one at 1024^2 x 1 (separate passes, --pmc with --kernel-trace only)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/pmc_groups
for cfg in "256 4" "1024 1"; do set -- $cfg
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_groups/$1x$2_p$i" -o p$i -- python "$GRAFT_REPO_ROOT/scripts/drive.py" --map-size $1 --cascades $2 --frames 81 --warmup 2) > gpurun_out/pmc_groups/$1x$2_p$i.log 2>&1
    echo "pmc $1x$2 pass $i rc=$?"
  done
done
python scripts/rocprof_summary.py gpurun_out/pmc_groups gpurun_out/pmc_groups_summary.txt; grep -E "k_tick_group.*(FETCH_SIZE|WRITE_SIZE)" gpurun_out/pmc_groups_summary.txt | cut -c1-200
