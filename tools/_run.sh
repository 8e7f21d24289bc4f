mkdir -p gpurun_out/r5
timeout 600 python tools/kv_group_lab.py 2>&1 | grep -v amdgpu > gpurun_out/r5/k2_kv_lab.txt; cut -c1-200 gpurun_out/r5/k2_kv_lab.txt
timeout 900 python -m pytest tests/test_gemm_configs.py -x -q -k "grouped" 2>&1 | tail -2
for i in 1 2 3; do timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; done
