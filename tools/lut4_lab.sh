#!/bin/bash
# TIMING-ONLY lab build (wrong results): the 256x256 half-tile-ring GEMM with the expansion a fused 4-bit loader would run on every
# weight fragment (gemm.hip, SDNQ_LAB_LUT4) -> build/libsdnq_hip_lut4.so.  Run the large GEMM shapes on both libraries:
#   tools/lut4_lab.sh && python tools/bench_gemm.py int8 all; SDNQ_HIP_LIB=$PWD/build/libsdnq_hip_lut4.so python tools/bench_gemm.py int8 all
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT/sdnq_amd/csrc"
mkdir -p "$ROOT/build"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-command-line-argument -DSDNQ_LAB_LUT4"
OBJS=()
for f in api rowquant gemm dequant quantize conv attention parallel; do
  X=""  # per-file flags as in sdnq_amd/csrc/build.sh
  [ $f = attention ] && X="-mllvm -amdgpu-mfma-vgpr-form"
  [ $f = rowquant ] && X="-DSDNQ_PRELOAD_ROWQUANT -mllvm -amdgpu-kernarg-preload-count=14"
  [ $f = gemm ] && X="-DSDNQ_PRELOAD_GEMM -mllvm -amdgpu-kernarg-preload-count=14"
  { [ $f = dequant ] || [ $f = conv ]; } && X="-mllvm -amdgpu-kernarg-preload-count=14"
  /opt/rocm/bin/hipcc $F $X -c $f.hip -o /tmp/lut4_$f.o & OBJS+=(/tmp/lut4_$f.o)
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wno-unused-command-line-argument -o "$ROOT/build/libsdnq_hip_lut4.so" "${OBJS[@]}"
echo built
