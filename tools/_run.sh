mkdir -p gpurun_out/r6/final
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/r6/final/pytest_gpu_serial.txt
cat gpurun_out/r6/final/pytest_gpu_serial.txt
