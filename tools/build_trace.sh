#!/bin/bash
# development build of the library with per-workgroup phase timestamps in the GEMM kernel -> build/libsdnq_hip_trace.so
# (use with SDNQ_HIP_LIB=$PWD/build/libsdnq_hip_trace.so python tools/trace_gemm.py)
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT/sdnq_amd/csrc"
mkdir -p "$ROOT/build"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-command-line-argument -DSDNQ_TRACE"
OBJS=()
for f in api rowquant gemm dequant quantize conv attention parallel; do
  X=""  # per-file flags as in sdnq_amd/csrc/build.sh
  [ $f = attention ] && X="-mllvm -amdgpu-mfma-vgpr-form"
  [ $f = rowquant ] && X="-DSDNQ_PRELOAD_ROWQUANT -mllvm -amdgpu-kernarg-preload-count=14"
  [ $f = gemm ] && X="-DSDNQ_PRELOAD_GEMM -mllvm -amdgpu-kernarg-preload-count=14"
  { [ $f = dequant ] || [ $f = conv ]; } && X="-mllvm -amdgpu-kernarg-preload-count=14"
  /opt/rocm/bin/hipcc $F $X -c $f.hip -o /tmp/trace_$f.o & OBJS+=(/tmp/trace_$f.o)
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wno-unused-command-line-argument -o "$ROOT/build/libsdnq_hip_trace.so" "${OBJS[@]}"
echo built
