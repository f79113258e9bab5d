#!/bin/bash
# SQ activity counters of the tick-pair kernel at 1024^2 x 4 (each pass its own run; --pmc only with --kernel-trace)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/pmc_pairs
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS" ; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_pairs/p$i" -o p$i -- python "$GRAFT_REPO_ROOT/scripts/drive.py" --map-size 1024 --cascades 4 --frames 21 --warmup 2) > gpurun_out/pmc_pairs/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python scripts/rocprof_summary.py gpurun_out/pmc_pairs gpurun_out/pmc_pairs_summary.txt; grep -E "k_tick_pair" gpurun_out/pmc_pairs_summary.txt | grep -v "^ " | cut -c1-170
