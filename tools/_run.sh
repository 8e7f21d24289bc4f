mkdir -p gpurun_out/r5
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_gpu.py -x -q -n 4 2>&1 | tail -4 > gpurun_out/r5/j1_pytest.txt
cat gpurun_out/r5/j1_pytest.txt
