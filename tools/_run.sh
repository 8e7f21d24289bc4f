# scratch command file of the build sessions: `gpurun -- 'bash tools/_run.sh'` (edited per experiment).  Last content: the closing measurement pack.
mkdir -p gpurun_out/r5/final2
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r5/final2/pytest_gpu_serial.txt
cat gpurun_out/r5/final2/pytest_gpu_serial.txt
# PMC traffic of the step's 443 scaled-mm launches (the set the bench's roofline replays): one-launch route off so the set is the same
SDNQ_HIP_FUSED_ROWQUANT=0 bash tools/pmc_step.sh r5/final2_pmc > gpurun_out/r5/final2/pmc.log 2>&1
cp gpurun_out/r5/final2_pmc/pmc_gemm_traffic.json gpurun_out/r5/final2/r05_pmc_gemm_traffic_linked.json
cp gpurun_out/r5/final2_pmc/pmc_rowquant_traffic.json gpurun_out/r5/final2/r05_pmc_rowquant_traffic_linked.json
bash tools/prof_bench.sh r5/final2_prof --steps 20 --warmup 3 > gpurun_out/r5/final2/prof.log 2>&1
timeout 900 python bench.py > gpurun_out/r5/final2/bench_sdxl_int8.json 2> gpurun_out/r5/final2/bench.err
for w in sdxl_fp8 sdxl_int8_dequant flux_int4_had flux_int8_svd sdxl_conv_int8 sdxl_attn_int8 linear_int8 sdxl_unet_all; do
timeout 900 python bench.py --workload $w --no-cpu-baseline > gpurun_out/r5/final2/bench_$w.json 2>> gpurun_out/r5/final2/bench.err
done
