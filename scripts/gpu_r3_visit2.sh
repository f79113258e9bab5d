#!/bin/bash
# round 3, visit 2: new tests (interop, compiled Water node), A/B of merged launches vs one launch per pass in one process
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3v2
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_interop.py tests/test_water_host.py tests/test_runtime_contract.py -m gpu -q > gpurun_out/r3v2/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3v2/pytest.log
tail -40 gpurun_out/r3v2/pytest.log
timeout 600 python scripts/ab_merged.py > gpurun_out/r3v2/ab_merged.txt 2>&1
cat gpurun_out/r3v2/ab_merged.txt
