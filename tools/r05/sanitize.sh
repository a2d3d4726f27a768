#!/bin/bash
# Sanitizer runs of libfpcc_host (CPU only; GPU sanitizers are not available on this pool):
#   1. tools/r05/pool_stress.cpp built together with rans_host.cpp under ThreadSanitizer and under AddressSanitizer + UBSan
#      (two contexts x 800 frames x 8 jobs = 12 800 flag-gated jobs each);
#   2. the Python CPU tests of the pool and of the frame pipeline against a sanitized build of the library
#      (FPCC_HOST_LIB=<path> makes fastpcc_amd._native load it; the sanitizer runtime is preloaded into python).
# Output: profiles/r05/sanitize.log
set -u
cd "$(dirname "$0")/../.."
OUT=profiles/r05/sanitize.log
SRC=fastpcc_amd/csrc/host/rans_host.cpp
FR=${1:-800}
: > $OUT
for kind in thread address,undefined; do   # (round 5, later: the row warmers of the 255-ary decoder are covered by the Python leg, test_rans_golden.py)
  tag=${kind%%,*}
  echo "== pool_stress under -fsanitize=$kind" | tee -a $OUT
  g++ -O1 -g -fno-omit-frame-pointer -fsanitize=$kind -std=c++17 -pthread -march=x86-64-v3 tools/r05/pool_stress.cpp $SRC -o /tmp/pool_$tag || exit 1
  TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1" ASAN_OPTIONS="detect_leaks=1" UBSAN_OPTIONS="print_stacktrace=1" \
    /tmp/pool_$tag $FR 2 2>&1 | tail -40 | tee -a $OUT
  echo "exit code ${PIPESTATUS[0]}" | tee -a $OUT
  echo "== python CPU tests against libfpcc_host built with -fsanitize=$kind" | tee -a $OUT
  g++ -O1 -g -fno-omit-frame-pointer -fsanitize=$kind -std=c++17 -fPIC -shared -pthread -march=x86-64-v3 -o /tmp/libfpcc_host_$tag.so $SRC || exit 1
  rt=$(g++ -print-file-name=lib${tag:0:1}san.so)
  [ "$tag" = thread ] && rt=$(g++ -print-file-name=libtsan.so) || rt=$(g++ -print-file-name=libasan.so)
  FPCC_HOST_LIB=/tmp/libfpcc_host_$tag.so LD_PRELOAD=$rt ASAN_OPTIONS="detect_leaks=0" \
    TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 ignore_noninstrumented_modules=1" \
    timeout 1500 python -m pytest tests/test_coder_pool.py tests/test_serving.py tests/test_rans_golden.py -q -x -p no:cacheprovider 2>&1 | tail -15 | tee -a $OUT
done
grep -c "WARNING: ThreadSanitizer\|ERROR: AddressSanitizer\|runtime error" $OUT | sed 's/^/sanitizer reports in the log: /' | tee -a $OUT
