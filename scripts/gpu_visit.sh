#!/bin/bash
# generic visit: full GPU suite, then whatever "$@" names (commands run with bash -c, logs under gpurun_out/$OUT)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=${OUT:-visit}; mkdir -p gpurun_out/$OUT; export TMPDIR=/tmp
if [ -z "$NO_TESTS" ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/$OUT/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$OUT/pytest.log
  tail -6 gpurun_out/$OUT/pytest.log
fi
i=0
for cmd in "$@"; do
  i=$((i+1)); echo "== $cmd" | tee gpurun_out/$OUT/cmd$i.log
  timeout 900 bash -c "$cmd" >> gpurun_out/$OUT/cmd$i.log 2>&1; echo "rc=$?" >> gpurun_out/$OUT/cmd$i.log
  tail -25 gpurun_out/$OUT/cmd$i.log
done
