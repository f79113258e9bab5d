#!/bin/bash
# memory-side traffic of the frame kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (they do not fit one: TCC has 4 slots,
# FETCH_SIZE takes 3 and WRITE_SIZE 2), --kernel-trace beside them and nothing else.   R=r03 bash scripts/gpu_pmc_traffic.sh
cd "$GRAFT_REPO_ROOT" || exit 1
R=${R:-r03}; O=gpurun_out/${R}_pmc; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for cfg in "1024 4 21" "2048 1 21" "1024 8 21" "256 4 81"; do
  set -- $cfg
  for ctr in FETCH_SIZE WRITE_SIZE; do
    d=$O/n$1x$2_$ctr
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d "$GRAFT_REPO_ROOT/$d" -o p -- python "$GRAFT_REPO_ROOT/scripts/drive.py" --map-size $1 --cascades $2 --frames $3 --warmup 2) > $d.log 2>&1
    echo "$cfg $ctr rc=$?"
  done
done
python scripts/rocprof_summary.py $O $O/summary.txt
grep -E "^## |FETCH_SIZE|WRITE_SIZE" $O/summary.txt | grep -E "^## |k_tick|k_pass" | cut -c1-170
