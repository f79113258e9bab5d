#!/bin/bash
# developer tool: libocean_waves with the HOST side of every translation unit under a sanitizer -- the device code is
# compiled as always (-fno-gpu-sanitize: GPU AddressSanitizer needs xnack+ code objects, which the pool does not run), so the kernels are the
# shipped ones and what is checked is the 1 800 lines of runtime / look-ahead / ring arithmetic and the group's worker threads.
#   scripts/build_sanitized.sh asan   -> godotoceanwaves_amd/csrc/build/variants/asan.so   (-fsanitize=address,undefined)
#   scripts/build_sanitized.sh tsan   -> .../variants/tsan.so                               (-fsanitize=thread)
# run:  scripts/run_sanitized.sh   (on a GPU box; the recipe and its output: profiles/r06_sanitizers.txt)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); C=$ROOT/godotoceanwaves_amd/csrc; kind=${1:-asan}
case $kind in
  asan) SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined";;
  tsan) SAN="-fsanitize=thread";;
  *) echo "asan | tsan"; exit 2;;
esac
out=$C/build/variants; mkdir -p $out/$kind
F="-g -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -fvisibility=hidden -Wno-unused-function -fno-omit-frame-pointer -fno-gpu-sanitize -shared-libsan $SAN"
hipcc -O3 $F -c $C/ow_frame.hip -o $out/$kind/ow_frame.o &
hipcc -O3 $F -ffp-contract=off -c $C/ow_spectrum.hip -o $out/$kind/ow_spectrum.o &
hipcc -O1 $F -c $C/ow_runtime.hip -o $out/$kind/ow_runtime.o &
hipcc -O3 $F -ffp-contract=off -c $C/ow_consumer.hip -o $out/$kind/ow_consumer.o &
hipcc -O1 $F -c $C/ow_group.hip -o $out/$kind/ow_group.o &
wait
hipcc --offload-arch=gfx950 -shared -shared-libsan $SAN -o $out/$kind.so $out/$kind/*.o
echo $out/$kind.so
