mkdir -p gpurun_out/r5
timeout 2400 python -m pytest tests -x -q -m gpu -n 4 2>&1 | tail -8 > gpurun_out/r5/aq5_pytest_all.txt
cat gpurun_out/r5/aq5_pytest_all.txt
for w in sdxl_int8 sdxl_fp8; do for on in 0 1; do
SDNQ_HIP_FUSED_ROWQUANT=$on timeout 600 python bench.py --workload $w --steps 20 --warmup 3 > gpurun_out/r5/aq5_${w}_$on.json 2> gpurun_out/r5/aq5_${w}_$on.err
done; done
for on in 0 1; do SDNQ_HIP_FUSED_ROWQUANT=$on timeout 600 python bench.py --launch eager --steps 20 --warmup 3 > gpurun_out/r5/aq5_eager_$on.json 2> gpurun_out/r5/aq5_eager_$on.err; done
for f in sdxl_int8_0 sdxl_int8_1 sdxl_fp8_0 sdxl_fp8_1 eager_0 eager_1; do python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r5/aq5_$f.json").read().strip().splitlines()[-1]); print("$f", d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"])
except Exception as e: print("$f", "ERR", e)
PY
done
