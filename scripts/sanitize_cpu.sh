#!/bin/bash
# CPU sanitizer job (CPU only -- GPU AddressSanitizer is not available on the MI355X pool):
#   1. the plain-C restatement of the path (oracle/chimera_oracle_c.c: the checker of the full-size GPU tests and the CPU baseline) rebuilt with
#      AddressSanitizer + UndefinedBehaviorSanitizer, tests/test_oracle_c.py run against that build;
#   2. the socket control plane (chimera_amd.parallel.Rendezvous / HostComm, the 'params' scheme) under Python's development mode
#      (-X dev: unclosed sockets and files, unawaited resources become errors through -W error::ResourceWarning).
# Leaves the ordinary build of the oracle behind.   bash scripts/sanitize_cpu.sh
set -e
cd "$(dirname "$0")/.."
trap 'make -s -C oracle clean; make -s -C oracle' EXIT
make -s -C oracle clean
make -s -C oracle CFLAGS="-O1 -g -fopenmp -fPIC -ffp-contract=off -Wall -Wextra -Wno-unused-parameter -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer"
nm -D oracle/libchimera_oracle_c.so | grep -q __asan_init || { echo "sanitize_cpu: the oracle was not built with the sanitizers"; exit 1; }
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
  python -m pytest tests/test_oracle_c.py -x -q -p no:cacheprovider
make -s -C oracle clean
make -s -C oracle
python -X dev -W error::ResourceWarning -m pytest tests/test_sharding_cpu.py -x -q -p no:cacheprovider -k "rendezvous or params or host_combination"
echo "sanitize_cpu: ok"
