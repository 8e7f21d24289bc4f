#!/bin/bash
# The closing measurement pack of a round (run on the GPU box from the repo root): the GPU suite, the PMC traffic of every launch set on
# the shipped library (copied into profiles/ FIRST so that the bench lines carry non-stale `traffic`), the default bench line with its
# CPU baseline, the step-window kernel stats, the bench line of every other workload.   usage: tools/final_pack.sh <out-dir under gpurun_out>
D=gpurun_out/${1:-r6/final}
mkdir -p $D
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > $D/pytest_gpu_serial.txt
cat $D/pytest_gpu_serial.txt
for set in sdxl sdxl_fp8 sdxl_dequant linear flux flux_svd; do
  bash tools/pmc_step.sh ${1:-r6/final}_pmc_$set $set > $D/pmc_$set.log 2>&1
  cp gpurun_out/${1:-r6/final}_pmc_$set/pmc_gemm_traffic.json profiles/r06_pmc_gemm_traffic_$set.json
  cp gpurun_out/${1:-r6/final}_pmc_$set/pmc_gemm_traffic.json $D/r06_pmc_gemm_traffic_$set.json
  [ $set = sdxl ] && cp gpurun_out/${1:-r6/final}_pmc_$set/pmc_rowquant_traffic.json $D/r06_pmc_rowquant_traffic_sdxl.json
done
timeout 900 python bench.py > $D/bench_sdxl_int8.json 2> $D/bench.err
tail -c 400 $D/bench_sdxl_int8.json
bash tools/prof_bench.sh ${1:-r6/final}_prof --steps 20 --warmup 3 > $D/prof.log 2>&1
cp gpurun_out/${1:-r6/final}_prof_window_kernels.csv gpurun_out/${1:-r6/final}_prof_window_summary.txt gpurun_out/${1:-r6/final}_prof_kernel_stats.csv gpurun_out/${1:-r6/final}_prof_bench.json $D/ 2>/dev/null
for w in sdxl_fp8 sdxl_int8_dequant sdxl_int4 flux_int4_had flux_int8_svd sdxl_conv_int8 sdxl_attn_int8 linear_int8 sdxl_unet_all; do
  timeout 900 python bench.py --workload $w --no-cpu-baseline > $D/bench_$w.json 2>> $D/bench.err
  python - <<PY
import json
try:
    d = json.loads(open("$D/bench_$w.json").read().strip().splitlines()[-1]); r = d.get("roofline") or {}
    print("$w", d["ms_per_step"], "frac", r.get("frac"), "traffic_stale", r.get("traffic_stale"))
except Exception as e: print("$w", "FAILED", e)
PY
done
for mode in graph capture eager; do
  echo "launch=$mode: $(python bench.py --steps 20 --warmup 3 --no-cpu-baseline --launch $mode 2>/dev/null | tail -1 | python -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])')" | tee -a $D/launch_modes.txt
done
