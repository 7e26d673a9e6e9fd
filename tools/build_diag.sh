#!/usr/bin/env bash
# DIAGNOSTIC build of the library (NOT the product): -DJCDF_DIAGNOSTIC adds the timing-only ablation forms of the W kernel
# (wrong results), the in-kernel cycle stamps, the register-staged predecessor kernels, the experiment kernels
# (k_keepalive), the optional eigensolver paths that were measured at parity (csrc/jcdf_sbr.hpp two-stage reduction, Q
# replay) and the JCDF_* variant environment variables.  Output: tools/_build/libjcdf_hip_diag.so; select it with
#   JCDF_LIB_PATH=tools/_build/libjcdf_hip_diag.so python tools/<tool>.py
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
PKG="$ROOT/juliachem.jl_amd"
mkdir -p "$ROOT/tools/_build"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -DJCDF_DIAGNOSTIC \
    -Wall -Wno-unused-function -I"$ROOT/include" -Wl,--version-script="$PKG/csrc/exports.map" \
    "$PKG/csrc/jcdf_api.hip" "$PKG/csrc/jcint_host.cpp" -o "$ROOT/tools/_build/libjcdf_hip_diag.so" "$@"
echo "built: $ROOT/tools/_build/libjcdf_hip_diag.so (diagnostic)"
