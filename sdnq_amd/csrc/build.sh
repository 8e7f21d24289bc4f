#!/bin/bash
# Build the C-ABI shared library for gfx950 (MI355X). hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${1:-$HERE/../libsdnq_hip.so}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="${SDNQ_EXTRA_FLAGS:-} --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-command-line-argument"
OBJ="$HERE/../../build/obj"
mkdir -p "$OBJ"
pids=()
for f in api rowquant gemm dequant quantize conv attention; do
  EXTRA=""
  # attention.hip: keep the MFMA accumulators in VGPRs (the softmax rescales / reads them with VALU every block; in AGPR form
  # the compiler moved 80 registers per 32-key block through v_accvgpr_read/write)
  [ "$f" = attention ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form"
  ( "$HIPCC" $FLAGS $EXTRA -c "$HERE/$f.hip" -o "$OBJ/$f.o" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -Wno-unused-command-line-argument -o "$OUT" "$OBJ"/api.o "$OBJ"/rowquant.o "$OBJ"/gemm.o "$OBJ"/dequant.o "$OBJ"/quantize.o "$OBJ"/conv.o "$OBJ"/attention.o
echo "built $OUT"
