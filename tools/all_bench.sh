# every bench.py workload once (no CPU baseline) + the eager / compiled launches of the headline -> gpurun_out/$1/bench_<workload>.json
OUT=gpurun_out/${1:-r3b}
mkdir -p $OUT
for w in sdxl_int8 sdxl_int8_dequant sdxl_fp8 sdxl_unet_all sdxl_conv_int8 flux_int4_had flux_int8_svd linear_int8 sdxl_attn_int8 flux_attn_int8; do
  python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_$w.json
  python - <<PY
import json
d=json.load(open("$OUT/bench_$w.json"))
print("$w", d["ms_per_step"], "ms", d["value"], d["unit"], d.get("roofline",{}).get("frac"))
PY
done
python bench.py --launch eager --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_sdxl_int8_eager.json
python bench.py --launch compile --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_sdxl_int8_compile.json
SDNQ_HIP_CACHE_WEIGHTS=0 python bench.py --workload flux_int4_had --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_flux_int4_had_percall.json
python -c "
import json
for n in ('sdxl_int8_eager','sdxl_int8_compile','flux_int4_had_percall'):
    print(n, json.load(open('$OUT/bench_%s.json' % n))['ms_per_step'])"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --tp --workload flux_int8_svd --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_flux_int8_svd_tp1.json
python -c "import json; d=json.load(open('$OUT/bench_flux_int8_svd_tp1.json')); print('tp1 flux_int8_svd', d['ms_per_step'], d.get('tp'))"
