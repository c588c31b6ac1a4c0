#!/bin/bash
# The HOST logic of the product (plan building, container tags, size bounds, header parsing, the CNN weight pack: llicti_amd/csrc/host_types.hpp,
# cnn_pack.hpp, host_plan.hpp -- no HIP in them) under AddressSanitizer + UBSan, built by g++ (GPU sanitizers are not available on the pool).
# Not collected by pytest; run from the repo root:  bash tests/sanitize_host.sh     (tests/test_host_cpu.py runs the same driver unsanitized)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/llicti_asan_host
mkdir -p "$OUT"
g++ -O1 -g -std=c++17 -Wall -Wextra -Wno-unused-function -Wno-unused-parameter -fsanitize=address,undefined -fno-sanitize-recover=all -fno-omit-frame-pointer \
    -o "$OUT/sanitize_host" "$ROOT/tests/sanitize_host.cpp"
ASAN_OPTIONS=detect_leaks=1 "$OUT/sanitize_host"
