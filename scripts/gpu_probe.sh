#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_surface_sampling.py tests/test_c_consumer.py -q 2>&1 | tail -8
