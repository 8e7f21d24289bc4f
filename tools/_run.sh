mkdir -p gpurun_out/r5
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "prefetch or conv_quantizer" 2>&1 | tail -5 > gpurun_out/r5/t24.txt
J='import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])["ms_per_step"])'
for rep in 1 2; do echo "sdxl_conv_int8: $(python bench.py --workload sdxl_conv_int8 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | python -c "$J" 2>&1 | tail -1)" >> gpurun_out/r5/t24.txt; done
cat gpurun_out/r5/t24.txt
