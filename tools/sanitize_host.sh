#!/bin/bash
# Sanitizer runs of the integration library's HOST code on the CPU (the pool has no GPU
# AddressSanitizer; the device is off in these runs): builds daala_amd/host twice into /tmp -
# AddressSanitizer + UndefinedBehaviorSanitizer, then ThreadSanitizer - and drives inter, lossless
# and multi-worker intra streams through both seams (tools/sanitize_host.py).  Needs the reference
# sources (dev container).  Expected output: only the reference's own `clz(0)` notes
# (src/filter.c:1829, src/generic_code.c:67).
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
make -C daala_amd/host B=/tmp/build_asan OPT="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -march=x86-64-v3 -ffp-contract=off" > /dev/null
make -C daala_amd/host B=/tmp/build_tsan OPT="-O1 -g -fsanitize=thread -fno-omit-frame-pointer -march=x86-64-v3 -ffp-contract=off" > /dev/null
echo "== AddressSanitizer + UndefinedBehaviorSanitizer"
LD_LIBRARY_PATH=$ROOT/daala_amd LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
  ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 OD_HIPENC_LIB=/tmp/build_asan/libdaala_hipenc.so HIPENC_MV_EPZS=2 \
  python3 tools/sanitize_host.py 2>&1 | grep -E "runtime error|ERROR: AddressSanitizer|SUMMARY|^ok|Traceback" | sed -e 's/:[0-9]*:[0-9]*: runtime/: runtime/' | sort | uniq -c | sort -rn
echo "== ThreadSanitizer"
LD_LIBRARY_PATH=$ROOT/daala_amd LD_PRELOAD=$(gcc -print-file-name=libtsan.so) TSAN_OPTIONS=report_signal_unsafe=0:exitcode=0 \
  OD_HIPENC_LIB=/tmp/build_tsan/libdaala_hipenc.so python3 tools/sanitize_host.py --threads 2>&1 | grep -E "WARNING: ThreadSanitizer|SUMMARY|^ok|Traceback" | sort | uniq -c
