#!/bin/bash
# Attempt to build the reference's own hash-grid kernel (core/nets/occnerf/gridencoder/src/gridencoder.cu + bindings.cpp)
# for gfx950 as `oracle/_ref/_ref_gridencoder.so`, from the sources where they lie under /root/reference, with this image's
# tools only: hipify-perl -> hipcc against the installed torch headers.  (Not the reference's setup.py / backend.py.)
#
# RESULT ON THIS IMAGE (ROCm 7.2.0, torch 2.10.0+rocm7.0): UNBUILDABLE.  The translated source fails at gridencoder.cu:331,
#     atomicAdd((__half2*)&grad_grid[index + c], v);      error: no matching function for call to 'atomicAdd'
# -- HIP has no atomicAdd(__half2*, __half2) overload (grep -rn half2 /opt/rocm/include/hip/amd_detail/amd_hip_atomic.h
# amd_hip_unsafe_atomics.h: nothing), and the call sits in a plain `if (std::is_same<...>)`, not an `if constexpr`, so the
# float instantiation needs it to resolve as well.  Making it compile would take a hand-written stand-in for a CUDA
# intrinsic the image lacks, which the build rules exclude; the encoder therefore stays pinned by the restatement in
# oracle/occnerf_oracle.c.  This script is kept so that the failure can be reproduced:
#     bash oracle/ref_build_attempt.sh      (build container only; writes only under oracle/_ref/, which is git-ignored)
set -u
HERE=$(cd "$(dirname "$0")" && pwd)
SRC=/root/reference/core/nets/occnerf/gridencoder/src
OUT=$HERE/_ref
[ -d "$SRC" ] || { echo "no reference checkout at $SRC"; exit 2; }
mkdir -p "$OUT"
T=$(python3 -c 'import torch, os; print(os.path.dirname(torch.__file__))')
hipify-perl "$SRC/gridencoder.cu" 2>"$OUT/hipify.log" | sed 's#ATen/cuda/HIPContext.h#ATen/hip/HIPContext.h#' > "$OUT/gridencoder.hip"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DTORCH_EXTENSION_NAME=_ref_gridencoder \
    -DTORCH_API_INCLUDE_EXTENSION_H -D__HIP_PLATFORM_AMD__=1 -DUSE_ROCM=1 -DHIPBLAS_V2 \
    -I"$SRC" -I"$T/include" -I"$T/include/torch/csrc/api/include" $(python3 -m pybind11 --includes) \
    "$OUT/gridencoder.hip" "$SRC/bindings.cpp" -L"$T/lib" -lc10 -lc10_hip -ltorch_cpu -ltorch_hip -ltorch -ltorch_python \
    -Wl,-rpath,"$T/lib" -o "$OUT/_ref_gridencoder.so" 2>"$OUT/build.log"
rc=$?
grep -E "error" -A3 "$OUT/build.log" | head -12
echo "hipcc exit code $rc"
exit $rc
