#!/bin/bash
# Build the HIP library (so the snapshot carries a fresh .so), then run a command on the MI355X box.
# usage: tools/gpu.sh <timeout-seconds> '<command>'
set -euo pipefail
cd "$(dirname "$0")/.."
python -c "from sdnq_amd import _lib; _lib.build()" >/dev/null
T="$1"; shift
exec /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
