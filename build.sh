#!/usr/bin/env bash
# Builds libjcdf_hip.so (gfx950 only) and the CPU oracle library, in-tree.
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
PKG="$ROOT/juliachem.jl_amd"
mkdir -p "$PKG/lib" "$ROOT/oracle/_build"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread \
    -Wall -Wno-unused-function \
    -I"$ROOT/include" -Wl,--version-script="$PKG/csrc/exports.map" \
    "$PKG/csrc/jcdf_api.hip" "$PKG/csrc/jcint_host.cpp" -o "$PKG/lib/libjcdf_hip.so" "$@"
gcc -O2 -fPIC -shared -o "$ROOT/oracle/_build/libjcdf_oracle.so" "$ROOT/oracle/c/jcdf_oracle.c"
# CPU baseline of bench.py (the reference's two CPU modes on the host BLAS found at run time); checker/bench code only
gcc -O2 -fPIC -shared -fopenmp -o "$ROOT/oracle/_build/libjcdf_cpu_baseline.so" "$ROOT/oracle/c/jcdf_cpu_baseline.c" -ldl
echo "built: $PKG/lib/libjcdf_hip.so  $ROOT/oracle/_build/libjcdf_oracle.so  $ROOT/oracle/_build/libjcdf_cpu_baseline.so"
# The DIAGNOSTIC build of the same sources (tools/_build/libjcdf_hip_diag.so: ablation forms, two-stage reduction, Q replay — never
# the product): built here so that the driver's GPU suite runs tests/test_diagnostic_paths.py against it in a child process
# (tests/test_diag_build_gpu.py).  JCDF_SKIP_DIAG_BUILD=1 skips it (quick edit-compile loops).
if [ "${JCDF_SKIP_DIAG_BUILD:-0}" != "1" ]; then
    bash "$ROOT/tools/build_diag.sh" "$@"
fi
