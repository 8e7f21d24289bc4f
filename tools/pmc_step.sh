#!/bin/bash
# HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes as MI355X_MICROARCH.md prescribes) + kernel-trace stats of
# the SDXL step's launch set -> gpurun_out/<tag>/ ; run on the GPU box from the repo root.   usage: tools/pmc_step.sh <tag>
TAG=$1
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o p --output-format csv -- python "$OLDPWD/tools/pmc_shapes.py" sdxl 1 linked > "$OUT/fetch.log" 2>&1)
(cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o p --output-format csv -- python "$OLDPWD/tools/pmc_shapes.py" sdxl 1 linked > "$OUT/write.log" 2>&1)
python tools/pmc_traffic.py "$OUT/fetch" "$OUT/write" gemm_kernel > "$OUT/pmc_gemm_traffic.json"
python tools/pmc_traffic.py "$OUT/fetch" "$OUT/write" rowquant_kernel > "$OUT/pmc_rowquant_traffic.json"
tail -8 "$OUT/pmc_gemm_traffic.json"
find "$OUT" -name "*.csv" -size +300k -delete
