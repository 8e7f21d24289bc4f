#!/bin/bash
# rocprofv3 counter passes over one GEMM shape / tile / data fill -> gpurun_out/<tag>/ (run on the GPU box from the repo root)
# usage: tools/pmc_gemm.sh <tag> M N K <tile> <data>
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE GRBM_COUNT" \
         "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TD_TD_BUSY_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $C -d "$OUT/p$i" -o p --output-format csv -- python "$OLDPWD/tools/run_one_gemm.py" "$@" > "$OUT/run$i.log" 2>&1)
done
python tools/pmc_csv.py "$OUT" gemm_kernel > "$OUT/summary.txt" 2>&1
cat "$OUT/run1.log" | tail -1
cat "$OUT/summary.txt"
find "$OUT" -name "*.csv" -size +200k -delete
