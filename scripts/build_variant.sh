#!/bin/bash
# developer tool: build libocean_waves with extra -D flags into godotoceanwaves_amd/csrc/build/variants/<name>.so
#   scripts/build_variant.sh map1 -DOW_P1C_MAP=1 ;  OCEAN_WAVES_LIB=.../variants/map1.so python scripts/mode_bench.py ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); C=$ROOT/godotoceanwaves_amd/csrc; name=$1; shift
out=$C/build/variants; mkdir -p $out/$name
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -fvisibility=hidden -Wno-unused-function"
hipcc $F "$@" -c $C/ow_frame.hip -o $out/$name/ow_frame.o &
hipcc $F "$@" -ffp-contract=off -c $C/ow_spectrum.hip -o $out/$name/ow_spectrum.o &
hipcc $F "$@" -c $C/ow_runtime.hip -o $out/$name/ow_runtime.o &
hipcc $F "$@" -ffp-contract=off -c $C/ow_consumer.hip -o $out/$name/ow_consumer.o &
hipcc $F "$@" -c $C/ow_group.hip -o $out/$name/ow_group.o &
wait
hipcc --offload-arch=gfx950 -shared -o $out/$name.so $out/$name/*.o
echo $out/$name.so
