#!/bin/bash
# developer tool: libocean_waves with its HOST logic under a sanitizer.  The two host-only translation units -- ow_runtime.hip
# (the runtime: look-ahead queue, scratch ring, run-after-run, 1 900 lines) and ow_group.hip (a worker thread per shard, mutex / condition-variable
# hand-offs) -- are plain C++ over the HIP runtime API and are compiled here with g++ and GCC's sanitizer runtimes; the three units that hold kernels
# come from the normal hipcc build, uninstrumented: the kernels are the shipped ones.
# (Why not clang's runtime: ROCm's libclang_rt.asan intercepts hsa_amd_memory_pool_allocate for GPU ASan, and without xnack+ code objects -- which this
#  pool does not run -- the first hipMalloc dies in that interceptor: "AddressSanitizer: out of memory: allocator is trying to allocate 0x400000 bytes",
#  profiles/r06_sanitizers.txt.  GCC's libasan has no such interceptors.)
#   scripts/build_sanitized.sh asan   -> godotoceanwaves_amd/csrc/build/variants/asan.so   (-fsanitize=address,undefined)
#   scripts/build_sanitized.sh tsan   -> .../variants/tsan.so                               (-fsanitize=thread)
# run:  scripts/run_sanitized.sh   (on a GPU box)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); C=$ROOT/godotoceanwaves_amd/csrc; kind=${1:-asan}
case $kind in
  asan) SAN="-fsanitize=address,undefined";;
  tsan) SAN="-fsanitize=thread";;
  *) echo "asan | tsan"; exit 2;;
esac
python -c "import sys; sys.path.insert(0, '$ROOT'); from godotoceanwaves_amd import build; build.build_library()" > /dev/null   # the hipcc objects of the kernel units
out=$C/build/variants; mkdir -p $out/$kind
F="-x c++ -std=c++17 -O1 -g -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -fno-omit-frame-pointer -fvisibility=hidden -Wall -Wno-unused-function -Wno-unknown-pragmas $SAN"
g++ $F -c $C/ow_runtime.hip -o $out/$kind/ow_runtime.o &
g++ $F -c $C/ow_group.hip -o $out/$kind/ow_group.o &
wait
g++ -shared $SAN -o $out/$kind.so $out/$kind/ow_runtime.o $out/$kind/ow_group.o $C/build/ow_frame.o $C/build/ow_spectrum.o $C/build/ow_consumer.o \
    -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -lpthread
echo $out/$kind.so
