#!/bin/bash
# Runs the host side of libocean_waves under sanitizers on a GPU box; builds come from scripts/build_sanitized.sh (they travel
# with the snapshot): ow_runtime.hip / ow_group.hip compiled by g++ with GCC's ASan + UBSan / TSan, the kernel units by hipcc as shipped.   OUT=name bash scripts/run_sanitized.sh  -> gpurun_out/$OUT/sanitizers.txt
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=${OUT:-r06_san}; O=gpurun_out/$OUT; mkdir -p $O; export TMPDIR=/tmp
ASAN_RT=$(gcc -print-file-name=libasan.so); TSAN_RT=$(gcc -print-file-name=libtsan.so); V=$PWD/godotoceanwaves_amd/csrc/build/variants
R=$O/sanitizers.txt; : > $R
say() { echo "$@" | tee -a $R; }
export OW_ASSUME_GPU=1
if [ -z "$ONLY_TSAN" ]; then
# ---- AddressSanitizer + UndefinedBehaviorSanitizer: the runtime (look-ahead queue, scratch ring, run-after-run), the group, readback, interop ----
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=0:abort_on_error=0:print_summary=1:log_path=$PWD/$O/asan_report
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0:log_path=$PWD/$O/ubsan_report
ASAN_TESTS="tests/test_lookahead.py tests/test_run_after_run.py tests/test_spectrum_resident.py tests/test_readback.py tests/test_tick_groups.py::test_tick_pairs_equal_one_launch_per_pass tests/test_tick_groups.py::test_group_depth_follows_the_run_and_the_scratch_grows_on_first_use"
say "== ASan + UBSan (host side; device code unchanged): LD_PRELOAD=$ASAN_RT OCEAN_WAVES_LIB=$V/asan.so ASAN_OPTIONS=$ASAN_OPTIONS"
say "   python -m pytest $ASAN_TESTS -m gpu -q"
LD_PRELOAD=$ASAN_RT OCEAN_WAVES_LIB=$V/asan.so timeout 2400 python -m pytest $ASAN_TESTS -m gpu -q --timeout 1200 -p no:cacheprovider > $O/asan_pytest.log 2>&1
LD_PRELOAD=$ASAN_RT OCEAN_WAVES_LIB=$V/asan.so timeout 1200 python -m pytest tests/test_two_chains.py -k "not callers_stream" -m gpu -q --timeout 600 -p no:cacheprovider > $O/asan_pytest_chains.log 2>&1; say "   tests/test_two_chains.py -k 'not callers_stream' (that one imports torch): $(tail -1 $O/asan_pytest_chains.log)"
LD_PRELOAD=$ASAN_RT OCEAN_WAVES_LIB=$V/asan.so timeout 1200 python -m pytest tests/test_group.py -k "not c4_exact_shape" -m gpu -q --timeout 600 -p no:cacheprovider > $O/asan_pytest_group.log 2>&1; say "   tests/test_group.py -k 'not c4_exact_shape' (that one spends its time in the OpenMP oracle): $(tail -1 $O/asan_pytest_group.log)"
say "   rc=$?  $(tail -1 $O/asan_pytest.log)"
say "   python -m pytest tests/test_runtime_contract.py tests/test_interop.py -m gpu -q      (these import torch into the sanitized process)"
LD_PRELOAD=$ASAN_RT OCEAN_WAVES_LIB=$V/asan.so timeout 1200 python -m pytest tests/test_runtime_contract.py tests/test_interop.py -m gpu -q --timeout 900 -p no:cacheprovider > $O/asan_pytest_torch.log 2>&1
say "   rc=$?  $(tail -1 $O/asan_pytest_torch.log)"
say "   python scripts/fuzz_schedule.py 3 901     (33 random schedules, merged against never-merging, bit for bit)"
LD_PRELOAD=$ASAN_RT OCEAN_WAVES_LIB=$V/asan.so timeout 2400 python scripts/fuzz_schedule.py 3 901 > $O/asan_fuzz.log 2>&1
say "   rc=$?  $(tail -1 $O/asan_fuzz.log)"
n=$(ls $O/asan_report* $O/ubsan_report* 2>/dev/null | wc -l)
say "   sanitizer report files: $n"
for f in $O/asan_report* $O/ubsan_report*; do [ -f "$f" ] && { say "--- $f"; grep -E "ERROR|SUMMARY|runtime error|#[0-9] " "$f" | grep -v "python3\|libpython" | head -40 | tee -a $R; }; done
fi
# ---- ThreadSanitizer (GCC 11's runtime does not cope with this kernel's address-space randomisation -- "FATAL: ThreadSanitizer: unexpected memory
# mapping" -- so the processes run under `setarch -R`): the group's worker threads (ow_group.hip: one worker per shard, mutex / condition-variable hand-offs) from a compiled C host ----
export TSAN_OPTIONS=halt_on_error=0:second_deadlock_stack=1:log_path=$PWD/$O/tsan_report:ignore_noninstrumented_modules=1
say "== TSan: examples/multi_gpu_host.c (compiled C99 host, eight shards, every shard through the peer path, overlapped gathers) against $V/tsan.so"
gcc -O1 -g -std=c99 -fsanitize=thread -Iinclude examples/multi_gpu_host.c -o /tmp/multi_gpu_host_tsan $V/tsan.so -Wl,-rpath,$V -Wl,-rpath-link,/opt/rocm/lib -lm >> $R 2>&1
timeout 900 setarch $(uname -m) -R /tmp/multi_gpu_host_tsan 512 1 120 8 0,0,0,0,0,0,0,0 peer > $O/tsan_host.log 2>&1
say "   rc=$?  $(tail -1 $O/tsan_host.log | cut -c1-200)"
say "   python -m pytest tests/test_group.py -m gpu -q -k 'not c4_exact_shape'  under LD_PRELOAD=$TSAN_RT (python itself uninstrumented: ignore_noninstrumented_modules=1; the C4 test spends its time in the OpenMP oracle)"
LD_PRELOAD=$TSAN_RT OCEAN_WAVES_LIB=$V/tsan.so timeout 1500 setarch $(uname -m) -R python -m pytest tests/test_group.py -m gpu -q -k "not c4_exact_shape" --timeout 600 -p no:cacheprovider > $O/tsan_pytest.log 2>&1
say "   rc=$?  $(tail -1 $O/tsan_pytest.log)"
n=$(ls $O/tsan_report* 2>/dev/null | wc -l)
say "   TSan report files: $n"
for f in $O/tsan_report*; do [ -f "$f" ] && { say "--- $f"; grep -E "WARNING|SUMMARY|#[0-9] " "$f" | head -60 | tee -a $R; }; done
tail -60 $R
