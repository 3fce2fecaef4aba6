#!/bin/bash
# The host simulation of the device algorithms under MemorySanitizer (tests/host_sim/msan_main.cpp): a use of a value that the device
# source reads before writing it (private arrays, the LDS slot incl. psel / pad, table rows) aborts the run with its origin.
# usage: bash tools/msan_host_sim.sh [out.txt]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT/tests/host_sim
/opt/rocm/lib/llvm/bin/clang++ -fsanitize=memory -fsanitize-memory-track-origins=2 -fno-omit-frame-pointer -O0 -g -std=c++17 -DC12381_CHECK_BOUNDS -pthread \
    -Wno-unused-value msan_main.cpp -o /tmp/msan_sim
MSAN_OPTIONS=halt_on_error=1 /tmp/msan_sim 2>&1 | tee ${1:-/dev/stdout}
