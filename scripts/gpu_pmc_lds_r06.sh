cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_pmc_lds; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for cfg in "1024 4" "2048 1"; do set -- $cfg
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL -d "$GRAFT_REPO_ROOT/$O/n$1x$2" -o p -- python "$GRAFT_REPO_ROOT/scripts/drive.py" --map-size $1 --cascades $2 --frames 21 --warmup 2 --single-stream) > $O/n$1x$2.log 2>&1; echo "$cfg rc=$?"
done
python scripts/rocprof_summary.py $O $O/summary.txt; grep -E "^## |k_tick_pair" $O/summary.txt | cut -c1-150; find $O -name "*.db" -delete
