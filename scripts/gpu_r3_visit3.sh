#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3v3
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_interop.py -m gpu -q > gpurun_out/r3v3/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3v3/pytest.log
tail -30 gpurun_out/r3v3/pytest.log
timeout 600 python scripts/placement_probe.py 1024:4 > gpurun_out/r3v3/placement_1024x4.txt 2>&1
cat gpurun_out/r3v3/placement_1024x4.txt
