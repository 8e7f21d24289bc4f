mkdir -p gpurun_out/r5
J='import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])["ms_per_step"])'
( for wl in sdxl_int8 ; do
B="python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline"
echo "$wl prefetch off: $(SDNQ_HIP_PREFETCH_NEXT=0 $B 2>&1 | python -c "$J" 2>&1 | tail -1)"
echo "$wl prefetch on : $($B 2>&1 | python -c "$J" 2>&1 | tail -1)"
done
for wl in flux_int4_had flux_int8_svd; do
B="python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline"
echo "$wl prefetch off: $(SDNQ_HIP_PREFETCH_NEXT=0 $B 2>&1 | python -c "$J" 2>&1 | tail -1)"
echo "$wl prefetch on : $($B 2>&1 | python -c "$J" 2>&1 | tail -1)"
done ) > gpurun_out/r5/t21_prefetch_product.txt 2>&1
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 >> gpurun_out/r5/t21_prefetch_product.txt
cat gpurun_out/r5/t21_prefetch_product.txt
