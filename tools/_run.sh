mkdir -p gpurun_out/r6
for set in sdxl_dequant linear; do
  bash tools/pmc_step.sh r6/final4_pmc_$set $set > gpurun_out/r6/pmc_$set.log 2>&1
  tail -12 gpurun_out/r6/pmc_$set.log
  cp gpurun_out/r6/final4_pmc_$set/pmc_gemm_traffic.json profiles/r06_pmc_gemm_traffic_$set.json
  cp gpurun_out/r6/final4_pmc_$set/pmc_gemm_traffic.json gpurun_out/r6/r06_pmc_gemm_traffic_$set.json
done
for w in sdxl_int8_dequant linear_int8; do
  timeout 900 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r6/bench4_$w.json
  python -c "
import json; d=json.load(open('gpurun_out/r6/bench4_$w.json')); r=d['roofline']; print('$w', d['ms_per_step'], r['frac'], r['traffic'], r['algorithmic_bytes_per_launch'], r['traffic_stale'])"
done
