for i in 1 2 3; do
SDNQ_HIP_CONV_PREFETCH=0 timeout 900 python bench.py --workload sdxl_conv_int8 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('conv off', d['ms_per_step'])"
timeout 900 python bench.py --workload sdxl_conv_int8 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('conv on', d['ms_per_step'])"
done
SDNQ_HIP_CONV_PREFETCH=0 timeout 900 python bench.py --workload sdxl_unet_all --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('all off', d['ms_per_step'])"
timeout 900 python bench.py --workload sdxl_unet_all --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('all on', d['ms_per_step'])"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -n 4 -k conv 2>&1 | tail -2
