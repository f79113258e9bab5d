#!/bin/bash
# round 6: what do the waves of the pair kernels WAIT for (1024^2 x 4 against 2048^2, whole launches on one stream)?  SQ wait / level counters, separate passes.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_pmc_wait; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
rocprofv3-avail list 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $O/sq_counters.txt; wc -l $O/sq_counters.txt
for cfg in "1024 4 21" "2048 1 21"; do
  set -- $cfg
  i=0
  for ctrs in "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_EXP_GDS SQ_ACTIVE_INST_FLAT SQ_WAVE_CYCLES" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_WAVE_CYCLES" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INSTS_SALU SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES" "SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES SQ_LEVEL_WAVES"; do
    i=$((i+1)); d=$O/n$1x$2_p$i
    (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $ctrs -d "$GRAFT_REPO_ROOT/$d" -o p -- python "$GRAFT_REPO_ROOT/scripts/drive.py" --map-size $1 --cascades $2 --frames $3 --warmup 2 --single-stream) > $d.log 2>&1
    echo "$cfg pass $i rc=$?"
  done
done
python scripts/rocprof_summary.py $O $O/summary.txt
grep -E "^## |k_tick_pair" $O/summary.txt | grep -v "^  " | cut -c1-170
find $O -name "*.db" -delete
