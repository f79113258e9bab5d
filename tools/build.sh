#!/bin/bash
# Builds the developer microbenchmarks (not part of the product) for gfx950: tools/kbench (1024), tools/kbench_2048,
# tools/kbench_small_256, tools/kbench_small_512.  The binaries are git-ignored; they travel to the GPU box with gpurun.
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -I godotoceanwaves_amd/csrc"
hipcc $F tools/kbench.hip -o tools/kbench &
hipcc $F -DKBENCH_N=2048 tools/kbench.hip -o tools/kbench_2048 &
hipcc $F -I tools -DKS_N=256 tools/kbench_small.hip -o tools/kbench_small_256 &
hipcc $F -I tools -DKS_N=512 tools/kbench_small.hip -o tools/kbench_small_512 &
hipcc $F tools/kbench_2048pair.hip -o tools/kbench_2048pair &
hipcc $F tools/cvtcheck.hip -o tools/cvtcheck &
wait
ls -la tools/kbench tools/kbench_2048 tools/kbench_small_256 tools/kbench_small_512
