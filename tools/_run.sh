mkdir -p gpurun_out/r5
timeout 1200 python tools/fuzz_fused.py 1 300 2>&1 | tail -4 > gpurun_out/r5/g1_fuzz_fused.txt
cat gpurun_out/r5/g1_fuzz_fused.txt
timeout 900 python -m pytest tests/test_fuzz_gpu.py -x -q -k one_launch 2>&1 | tail -3
