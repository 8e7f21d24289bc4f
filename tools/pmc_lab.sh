#!/bin/bash
# rocprofv3 counter passes over the GEMM lab executable -> gpurun_out/<tag>/summary.txt (run on the GPU box from the repo root)
# usage: tools/pmc_lab.sh <tag> <lab binary> <shapes> <ids> [env assignments...]
set -u
TAG=$1; BIN=$PWD/$2; SH=$3; IDS=$4; shift 4
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for e in "$@"; do export "$e"; done
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TD_TD_BUSY_sum" \
         "TA_BUSY_avr TD_BUSY_avr TCP_TA_TCP_STATE_READ_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $C -d "$OUT/p$i" -o p --output-format csv -- "$BIN" "$SH" "$IDS" 2 > "$OUT/run$i.log" 2>&1)
done
python tools/pmc_csv.py "$OUT" gemm_kernel > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
find "$OUT" -name "*.csv" -size +200k -delete
