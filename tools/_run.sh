mkdir -p gpurun_out/r6
bash tools/prof_bench.sh r6/prof_sdxl_int8 --steps 20 --warmup 3 > gpurun_out/r6/prof_sdxl_int8.log 2>&1
tail -25 gpurun_out/r6/prof_sdxl_int8.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r6/bench_b.json 2> gpurun_out/r6/bench_b.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6/bench_b.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["frac"])
for r in d["roofline"].get("per_shape", []): print(r)
PY
for set in sdxl sdxl_fp8 flux flux_svd; do
bash tools/pmc_step.sh r6/pmc_$set $set > gpurun_out/r6/pmc_$set.log 2>&1
cp gpurun_out/r6/pmc_$set/pmc_gemm_traffic.json gpurun_out/r6/r06_pmc_gemm_traffic_$set.json
tail -6 gpurun_out/r6/pmc_$set.log
done
O=gpurun_out/r6/launch_modes_2.txt; : > $O
for i in 1 2; do for mode in graph capture eager; do
ms=$(python bench.py --steps 20 --warmup 3 --no-cpu-baseline --launch $mode 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')
echo "launch=$mode: $ms" | tee -a $O
done; done
