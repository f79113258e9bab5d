#!/bin/bash
# PMC counter passes (each its own run, --pmc never combined with tracing other than kernel-trace)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/pmc
N=${1:-1024}; C=${2:-4}; TAG=${3:-v0}
cd /tmp
rocprofv3 -L > "$GRAFT_REPO_ROOT/gpurun_out/pmc/counters_list.txt" 2>&1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "GRBM_GUI_ACTIVE GRBM_COUNT" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_INSTS_VALU_TRANS SQ_INSTS_FLAT" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d "$GRAFT_REPO_ROOT/gpurun_out/pmc/${TAG}_p$i" -o p$i -- python "$GRAFT_REPO_ROOT/scripts/drive.py" --map-size $N --cascades $C --frames 10 --warmup 2 > "$GRAFT_REPO_ROOT/gpurun_out/pmc/${TAG}_p$i.log" 2>&1
  echo "pass $i rc=$?"
done
cd "$GRAFT_REPO_ROOT"; python scripts/rocprof_summary.py gpurun_out/pmc gpurun_out/pmc/${TAG}_summary.txt; grep -v "^#" gpurun_out/pmc/${TAG}_summary.txt | grep -E "k_pass|##" | head -80
