#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 800 python -m pytest tests -m gpu -q 2>&1 | tail -15
