mkdir -p gpurun_out/r5
timeout 2400 python -m pytest tests -x -q -m gpu -n 4 2>&1 | tail -6 > gpurun_out/r5/b2_pytest_all.txt
cat gpurun_out/r5/b2_pytest_all.txt
for i in 1 2; do
SDNQ_HIP_LIB=$PWD/sdnq_amd/libsdnq_hip_ab.so timeout 900 python bench.py --workload flux_int8_svd --steps 10 --warmup 2 > gpurun_out/r5/b2_flux8_old$i.json 2> gpurun_out/r5/b2_flux8_old$i.err
timeout 900 python bench.py --workload flux_int8_svd --steps 10 --warmup 2 > gpurun_out/r5/b2_flux8_new$i.json 2> gpurun_out/r5/b2_flux8_new$i.err
done
for f in flux8_old1 flux8_new1 flux8_old2 flux8_new2; do python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r5/b2_$f.json").read().strip().splitlines()[-1]); print("$f", d["ms_per_step"])
except Exception as e: print("$f", "ERR", e)
PY
done
