#!/bin/bash
# development build of the library with per-workgroup phase timestamps in the GEMM kernel
set -euo pipefail
cd "$(dirname "$0")/../sdnq_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-command-line-argument -DSDNQ_TRACE"
for f in api rowquant gemm dequant quantize conv; do /opt/rocm/bin/hipcc $F -c $f.hip -o /tmp/trace_$f.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wno-unused-command-line-argument -o "$(dirname "$0")/../../build/libsdnq_hip_trace.so" /tmp/trace_api.o /tmp/trace_rowquant.o /tmp/trace_gemm.o /tmp/trace_dequant.o /tmp/trace_quantize.o /tmp/trace_conv.o
echo built
