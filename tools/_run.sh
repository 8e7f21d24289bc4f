mkdir -p gpurun_out/r5
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -n 4 -k "svd or flux or cfg5 or one_call" 2>&1 | tail -3
for i in 1 2; do
SDNQ_HIP_OVERLAP_LOWRANK=0 timeout 900 python bench.py --workload flux_int8_svd --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r5/i1_svd_off$i.json 2> gpurun_out/r5/i1_err.txt
timeout 900 python bench.py --workload flux_int8_svd --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r5/i1_svd_on$i.json 2>> gpurun_out/r5/i1_err.txt
done
SDNQ_HIP_OVERLAP_LOWRANK=0 timeout 900 python bench.py --workload flux_int8_svd --launch eager --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r5/i1_svd_eager_off.json 2>> gpurun_out/r5/i1_err.txt
timeout 900 python bench.py --workload flux_int8_svd --launch eager --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r5/i1_svd_eager_on.json 2>> gpurun_out/r5/i1_err.txt
for f in svd_off1 svd_on1 svd_off2 svd_on2 svd_eager_off svd_eager_on; do python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r5/i1_$f.json").read().strip().splitlines()[-1]); print("$f", d["ms_per_step"])
except Exception as e: print("$f", "ERR", e)
PY
done
tail -3 gpurun_out/r5/i1_err.txt
