mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gemm_configs.py -x -q -k "fused_dequant" 2>&1 | tail -4 > gpurun_out/r5/c1_pytest.txt
cat gpurun_out/r5/c1_pytest.txt
TUNE_CANDS="5,6" TUNE_ALL=1 timeout 2400 python tools/tune_tiles_in_step.py sdxl_int8_dequant 20 > gpurun_out/r5/c1_tune_dequant.txt 2>&1
cat gpurun_out/r5/c1_tune_dequant.txt
