#!/bin/bash
# round-2 profiling visit: GPU suite, default bench line, rocprofv3 kernel-trace of the same command, kernel traces of the other
# configs, PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) for 1024^2 x 4 and 2048^2 x 1
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest.log; tail -4 gpurun_out/pytest.log
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-400
rm -rf gpurun_out/prof_r02
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r02" -o r02 -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline) > gpurun_out/prof_r02.log 2>&1
python scripts/rocprof_summary.py gpurun_out/prof_r02 gpurun_out/prof_r02_summary.txt; head -8 gpurun_out/prof_r02_summary.txt | cut -c1-160
for cfg in "256 4" "1024 1" "1024 8" "2048 4"; do set -- $cfg
  rm -rf gpurun_out/prof_r02_$1x$2
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r02_$1x$2" -o t -- python "$GRAFT_REPO_ROOT/scripts/drive.py" --map-size $1 --cascades $2 --frames 400 --warmup 100) > gpurun_out/prof_r02_$1x$2.log 2>&1
  python scripts/rocprof_summary.py gpurun_out/prof_r02_$1x$2 gpurun_out/prof_r02_$1x$2_summary.txt; echo "== $1 x $2"; head -5 gpurun_out/prof_r02_$1x$2_summary.txt | cut -c1-150
done
mkdir -p gpurun_out/pmc_r02
for cfg in "1024 4" "2048 1"; do set -- $cfg
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_r02/$1x$2_p$i" -o p$i -- python "$GRAFT_REPO_ROOT/scripts/drive.py" --map-size $1 --cascades $2 --frames 10 --warmup 2) > gpurun_out/pmc_r02/$1x$2_p$i.log 2>&1
    echo "pmc $1x$2 pass $i rc=$?"
  done
done
python scripts/rocprof_summary.py gpurun_out/pmc_r02 gpurun_out/pmc_r02_summary.txt; grep -E "k_pass.*(FETCH_SIZE|WRITE_SIZE)" gpurun_out/pmc_r02_summary.txt | cut -c1-170
