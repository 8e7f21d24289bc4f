mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "f16mm or float16" 2>&1 | grep -v "^  File\|Extension modules" | tail -25 | tee gpurun_out/r6/pytest_f16.txt
