#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_borrowed_resources.py -m gpu -q -x 2>&1 | tail -25
