mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_parallel_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/r5/t25.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "load_sdnq_model" 2>&1 | tail -5 >> gpurun_out/r5/t25.txt
cat gpurun_out/r5/t25.txt
