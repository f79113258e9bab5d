#!/bin/bash
# LDS activity / bank-conflict counters of the frame kernels under the slot maps of ow_device.h lds_slot (one rocprofv3 pass each, --kernel-trace
# beside the counters and nothing else); variants built with scripts/build_variant.sh mapN -DOW_LDS_SLOT_MAP=N
cd "$GRAFT_REPO_ROOT" || exit 1
R=${R:-r03}; O=gpurun_out/${R}_pmc_lds; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/godotoceanwaves_amd/csrc/build/variants
for lib in ${LIBS:-default map0 map1}; do
  for cfg in "1024 4" "2048 1"; do
    set -- $cfg
    if [ $lib = default ]; then unset OCEAN_WAVES_LIB; else export OCEAN_WAVES_LIB=$V/$lib.so; [ -f $OCEAN_WAVES_LIB ] || continue; fi
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES -d "$GRAFT_REPO_ROOT/$O/${lib}_n$1x$2" -o p -- python "$GRAFT_REPO_ROOT/scripts/drive.py" --map-size $1 --cascades $2 --frames 21 --warmup 2) > $O/${lib}_n$1x$2.log 2>&1
    echo "$lib $cfg rc=$?"
  done
done
unset OCEAN_WAVES_LIB
python scripts/rocprof_summary.py $O $O/summary.txt
grep -E "^## |SQ_LDS_BANK_CONFLICT|SQ_LDS_IDX_ACTIVE|SQ_INSTS_LDS" $O/summary.txt | grep -E "^## |k_tick_pair|k_pass2c<2048|k_pass1c_split" | cut -c1-150
