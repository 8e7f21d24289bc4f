mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "packed_uint8mm" 2>&1 | tail -15 > gpurun_out/r5/m1_pytest.txt; cat gpurun_out/r5/m1_pytest.txt
for seed in 7 8; do timeout 900 python tools/fuzz_modes.py $seed 250 2>&1 | grep -v amdgpu | tail -3; done 2>&1 | cut -c1-220
