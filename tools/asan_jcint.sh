#!/usr/bin/env bash
# AddressSanitizer + UBSan run of the host integral engine (CPU build only; GPU sanitizers are not available on the pool).
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="${TMPDIR:-/tmp}/jcint_asan"
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -pthread -I"$ROOT/include" \
    "$ROOT/juliachem.jl_amd/csrc/jcint_host.cpp" "$ROOT/tools/asan_jcint_driver.cpp" -o "$OUT"
"$OUT"
