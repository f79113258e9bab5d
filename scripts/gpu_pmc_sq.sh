#!/bin/bash
# SQ activity of the pair kernels (how busy the vector ALUs, the LDS and the vector-memory issue are while a launch moves its bytes): rocprofv3 --kernel-trace --pmc, separate
# passes (no tracing domains beside the counters), scripts/drive.py --single-stream so that every launch is whole.   R=r06 bash scripts/gpu_pmc_sq.sh
cd "$GRAFT_REPO_ROOT" || exit 1
R=${R:-r06}; O=gpurun_out/${R}_pmc_sq; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for cfg in "1024 4 21" "2048 1 21"; do
  set -- $cfg
  i=0
  for ctrs in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE"; do
    i=$((i+1)); d=$O/n$1x$2_p$i
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctrs -d "$GRAFT_REPO_ROOT/$d" -o p -- python "$GRAFT_REPO_ROOT/scripts/drive.py" --map-size $1 --cascades $2 --frames $3 --warmup 2 ${DRIVE_FLAGS:---single-stream}) > $d.log 2>&1
    echo "$cfg pass $i rc=$?"
  done
done
python scripts/rocprof_summary.py $O $O/summary.txt
grep -E "^## |k_tick_pair" $O/summary.txt | cut -c1-170
find $O -name "*.db" -delete
