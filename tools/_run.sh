mkdir -p gpurun_out/r6
O=gpurun_out/r6/ks_in_step_3.txt; : > $O
run() { local name="$1"; shift; local ms; ms=$(env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])'); echo "$name: $ms" | tee -a $O; }
for i in 1 2 3; do
run "A  no tile 28                               " SDNQ_HIP_KSPLIT=0
run "E1 tile 28 for K=5120, prefetch behind last stage" SDNQ_HIP_KSPLIT=0 SDNQ_HIP_TILE_MAP=1024x1280x5120=28 SDNQ_HIP_KS_PF=1
run "E2 tile 28 for K=5120, prefetch behind prologue  " SDNQ_HIP_KSPLIT=0 SDNQ_HIP_TILE_MAP=1024x1280x5120=28 SDNQ_HIP_KS_PF=2
run "E0 tile 28 for K=5120, tile 28 prefetches nothing" SDNQ_HIP_KSPLIT=0 SDNQ_HIP_TILE_MAP=1024x1280x5120=28 SDNQ_HIP_KS_PF=0
done
timeout 600 python -m pytest tests/test_gemm_configs.py -x -q -k "28" 2>&1 | tail -2 | tee -a $O
SDNQ_HIP_KS_PF=2 timeout 600 python -m pytest tests/test_gemm_configs.py -x -q -k "28" 2>&1 | tail -2 | tee -a $O
