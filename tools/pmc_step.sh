#!/bin/bash
# HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes as MI355X_MICROARCH.md prescribes) of a step's GEMM launch set
# -> gpurun_out/<tag>/ ; run on the GPU box from the repo root.
# usage: tools/pmc_step.sh <tag> [sdxl|sdxl_fp8|flux|flux_svd] [linked]      (default: sdxl linked = the headline's launch set)
TAG=$1
WL=${2:-sdxl}
LINK=${3-linked}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
(cd /tmp && timeout 1500 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o p --output-format csv -- python "$OLDPWD/tools/pmc_shapes.py" $WL 1 $LINK > "$OUT/fetch.log" 2>&1)
(cd /tmp && timeout 1500 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o p --output-format csv -- python "$OLDPWD/tools/pmc_shapes.py" $WL 1 $LINK > "$OUT/write.log" 2>&1)
python tools/pmc_traffic.py "$OUT/fetch" "$OUT/write" gemm_kernel > "$OUT/pmc_gemm_traffic.json"
python tools/pmc_traffic.py "$OUT/fetch" "$OUT/write" rowquant_kernel > "$OUT/pmc_rowquant_traffic.json"
tail -8 "$OUT/pmc_gemm_traffic.json"
find "$OUT" -name "*.csv" -size +300k -delete
