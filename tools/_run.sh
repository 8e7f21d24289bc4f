mkdir -p gpurun_out/r5
timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -n 4 2>&1 | tail -4 > gpurun_out/r5/f1_pytest.txt
cat gpurun_out/r5/f1_pytest.txt
