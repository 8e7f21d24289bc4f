mkdir -p gpurun_out/r5/final
# PMC traffic of the step's GEMM launch set (the 443 scaled-mm launches the bench's roofline replays): fused route off so the set is the same
SDNQ_HIP_FUSED_ROWQUANT=0 bash tools/pmc_step.sh r5/final_pmc > gpurun_out/r5/final/pmc.log 2>&1
tail -12 gpurun_out/r5/final/pmc.log
cp gpurun_out/r5/final_pmc/pmc_gemm_traffic.json profiles/r05_pmc_gemm_traffic_linked.json
cp gpurun_out/r5/final_pmc/pmc_rowquant_traffic.json profiles/r05_pmc_rowquant_traffic_linked.json
cp profiles/r05_pmc_gemm_traffic_linked.json profiles/r05_pmc_rowquant_traffic_linked.json gpurun_out/r5/final/
# kernel stats of the default bench command
bash tools/prof_bench.sh r5/final_prof --steps 20 --warmup 3 > gpurun_out/r5/final/prof.log 2>&1
tail -16 gpurun_out/r5/final/prof.log
# the default bench line (with the CPU baseline), twice
timeout 900 python bench.py > gpurun_out/r5/final/bench_sdxl_int8.json 2> gpurun_out/r5/final/bench.err
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r5/final/bench_sdxl_int8_b.json 2>> gpurun_out/r5/final/bench.err
for w in sdxl_fp8 sdxl_int8_dequant flux_int4_had flux_int8_svd sdxl_conv_int8 sdxl_attn_int8 linear_int8; do
timeout 900 python bench.py --workload $w --no-cpu-baseline > gpurun_out/r5/final/bench_$w.json 2>> gpurun_out/r5/final/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5/final/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get("roofline",{})
        print(f.split("/")[-1], d["ms_per_step"], r.get("frac"), r.get("avg_launch_us"), r.get("traffic"), r.get("traffic_stale"), d["config"].get("one_launch_linears"))
    except Exception as e: print(f, "ERR", e)
PY
