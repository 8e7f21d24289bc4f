mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gemm_configs.py -x -q -k "tile_configuration and (28 or 29 or 30 or 31)" 2>&1 | tail -5 > gpurun_out/r5/t28_pytest.txt
timeout 600 python tools/tile_ab.py "1024,1280,1280;1024,1280,5120;4096,640,640;4096,640,2560" "-1,28,29,30,31" > gpurun_out/r5/t28_ab.txt 2>&1
TUNE_SHAPES="1024x1280x1280,1024x1280x5120,4096x640x640,4096x640x2560" TUNE_CANDS="28,29,30,31" timeout 1500 python tools/tune_tiles_in_step.py sdxl_int8 20 > gpurun_out/r5/t28_tune.txt 2>&1
timeout 900 python -m pytest tests/test_attention.py -x -q -m gpu 2>&1 | tail -3 >> gpurun_out/r5/t28_pytest.txt
cat gpurun_out/r5/t28_*.txt
