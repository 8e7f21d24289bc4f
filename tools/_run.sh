mkdir -p gpurun_out/r6
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^  File\|Extension modules" | tail -15 | tee gpurun_out/r6/pytest_gpu_fp.txt
for mode in graph capture eager; do
  timeout 600 python bench.py --launch $mode --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r6/bench_fp_$mode.json
  python - <<PY
import json; d=json.load(open("gpurun_out/r6/bench_fp_$mode.json")); print("launch=$mode:", d["ms_per_step"])
PY
done
SDNQ_HIP_FAST_PLANS=0 timeout 600 python bench.py --launch eager --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r6/bench_nofp_eager.json
python -c "
import json; d=json.load(open('gpurun_out/r6/bench_nofp_eager.json')); print('launch=eager SDNQ_HIP_FAST_PLANS=0:', d['ms_per_step'])"
