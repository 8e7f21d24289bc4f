mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests/test_gemm_w4.py tests/test_cabi.py -x -q 2>&1 | grep -v "^  File\|Extension modules" | tail -5 | tee gpurun_out/r6/pytest_w4.txt
for i in 1 2; do for cfg in "1 1" "0 1" "0 0"; do set -- $cfg
SDNQ_HIP_CACHE_WEIGHTS=$1 SDNQ_HIP_FUSED_LUT4=$2 timeout 900 python bench.py --workload sdxl_int4 --no-cpu-baseline > gpurun_out/r6/bench_sdxl_int4_c$1_l$2.json 2> gpurun_out/r6/bench_sdxl_int4_c$1_l$2.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r6/bench_sdxl_int4_c$1_l$2.json").read().strip().splitlines()[-1])
print("sdxl_int4 CACHE_WEIGHTS=$1 FUSED_LUT4=$2", d["ms_per_step"], d["config"]["resident_weight_bytes"]["total"])
PY
done; done 2>&1 | tee gpurun_out/r6/sdxl_int4_modes.txt
