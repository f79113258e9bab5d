#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_surface_sampling.py tests/test_readback.py tests/test_golden.py -m gpu -q -x 2>&1 | tail -25
timeout 200 python scripts/pcie_rate.py 2>&1 | tee gpurun_out/pcie_rate.json | tail -5
