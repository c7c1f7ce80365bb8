#!/bin/bash
# The oracle (oracle/kkt_oracle.c, test infrastructure) under AddressSanitizer + UndefinedBehaviorSanitizer: an instrumented
# build in a scratch directory, loaded by the CPU tests through KKT_ORACLE_SO with the sanitizer runtimes preloaded into
# python.  UBSan halts on the first report, ASan aborts: a green run is a clean run.  CPU only (about 90 s).
#   scripts/sanitize_oracle.sh [pytest arguments]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$(mktemp -d)"
gcc -O1 -g -ffp-contract=off -fPIC -std=c99 -fsanitize=address,undefined -fno-omit-frame-pointer -shared \
    -o "$OUT/libkktoracle_asan.so" "$ROOT/oracle/kkt_oracle.c" -lm
cd "$ROOT"
KKT_ORACLE_SO="$OUT/libkktoracle_asan.so" \
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
python -m pytest tests/test_oracle_ldl.py tests/test_oracle_cones.py tests/test_structures_host.py tests/test_ipm_fixtures.py \
    -q -m "not gpu" -p no:cacheprovider "$@"
