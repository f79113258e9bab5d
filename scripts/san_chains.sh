cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06_s3; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_two_chains.py -m gpu -q --timeout 600 > $O/pytest_two_chains.log 2>&1; tail -3 $O/pytest_two_chains.log
ASAN_RT=$(gcc -print-file-name=libasan.so); V=$PWD/godotoceanwaves_amd/csrc/build/variants
export OW_ASSUME_GPU=1 ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=0:abort_on_error=0:print_summary=1:log_path=$PWD/$O/asan_report UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0:log_path=$PWD/$O/ubsan_report
LD_PRELOAD=$ASAN_RT OCEAN_WAVES_LIB=$V/asan.so timeout 1500 python -m pytest tests/test_two_chains.py tests/test_run_after_run.py -k "not callers_stream" -m gpu -q --timeout 1200 -p no:cacheprovider > $O/asan_pytest_chains.log 2>&1; echo "asan pytest rc=$?"; tail -2 $O/asan_pytest_chains.log
LD_PRELOAD=$ASAN_RT OCEAN_WAVES_LIB=$V/asan.so timeout 1500 python scripts/fuzz_schedule.py 1 977 > $O/asan_fuzz_chains.log 2>&1; echo "asan fuzz rc=$?"; tail -1 $O/asan_fuzz_chains.log
ls $O/asan_report* $O/ubsan_report* 2>/dev/null | wc -l
for f in $O/asan_report* $O/ubsan_report*; do [ -f "$f" ] && { echo "--- $f"; grep -E "ERROR|SUMMARY|runtime error|#[0-9] " "$f" | grep -v "python3\|libpython" | head -20; }; done
