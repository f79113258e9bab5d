#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/pmc_r02b
timeout 600 python scripts/gpu_err.py > gpurun_out/parity_margins_r02.txt 2>&1; cat gpurun_out/parity_margins_r02.txt | cut -c1-230
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_r02b/256x4_p$i" -o p$i -- python "$GRAFT_REPO_ROOT/scripts/drive.py" --map-size 256 --cascades 4 --frames 41 --warmup 41) > gpurun_out/pmc_r02b/256x4_p$i.log 2>&1; echo "pmc pass $i rc=$?"
done
python scripts/rocprof_summary.py gpurun_out/pmc_r02b gpurun_out/pmc_r02b_summary.txt; grep -E "k_tick.*(FETCH_SIZE|WRITE_SIZE)" gpurun_out/pmc_r02b_summary.txt | cut -c1-170
