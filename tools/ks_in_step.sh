#!/bin/bash
# In-step A/B of the K-split 64x80 tile (gemm_ks.hip, tile 28): same box, alternating arms, ms per step of bench.py (hipGraph replay).
# usage: tools/ks_in_step.sh <out-file> [rounds]
OUT="${1:-gpurun_out/r6/ks_in_step.txt}"; R="${2:-3}"
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
run() {  # name, env...
  local name="$1"; shift
  local ms
  ms=$(env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])')
  echo "$name: $ms" | tee -a "$OUT"
}
for i in $(seq 1 "$R"); do
  run "A heuristics without tile 28            " SDNQ_HIP_KSPLIT=0
  run "B tile 28 where preferred (K=5120 + rest)" SDNQ_HIP_KSPLIT=1
  run "C tile 28, one-launch w8a8 route off     " SDNQ_HIP_KSPLIT=1 SDNQ_HIP_FUSED_ROWQUANT=0
  run "D no tile 28, one-launch route off       " SDNQ_HIP_KSPLIT=0 SDNQ_HIP_FUSED_ROWQUANT=0
  run "E tile 28 for 1024x1280x5120 only        " SDNQ_HIP_KSPLIT=0 SDNQ_HIP_TILE_MAP=1024x1280x5120=28
done
