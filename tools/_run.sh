mkdir -p gpurun_out/r6
for wl in sdxl_fp8 sdxl_int8_dequant sdxl_int4 flux_int4_had flux_int8_svd sdxl_conv_int8 sdxl_unet_all; do
  for mode in graph eager; do
    timeout 900 python bench.py --workload $wl --launch $mode --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r6/bench_modes_${wl}_$mode.json
    python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r6/bench_modes_${wl}_$mode.json")); print("$wl launch=$mode:", d["ms_per_step"])
except Exception as e: print("$wl $mode failed", e)
PY
  done
done 2>&1 | tee gpurun_out/r6/launch_modes_all.txt
