mkdir -p gpurun_out/r2b
for w in sdxl_int8 sdxl_int8_dequant sdxl_fp8 sdxl_unet_all sdxl_conv_int8 flux_int4_had flux_int8_svd linear_int8; do
  python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2b/bench_$w.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r2b/bench_$w.json"))
print("$w", d["ms_per_step"], "ms", d["value"], d["unit"], d.get("roofline",{}).get("frac"))
PY
done
python bench.py --no-graph --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2b/bench_sdxl_int8_eager.json
python -c "import json;d=json.load(open('gpurun_out/r2b/bench_sdxl_int8_eager.json'));print('eager',d['ms_per_step'])"
