mkdir -p gpurun_out/r5
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r5/d1_base$i.json 2> gpurun_out/r5/d1_base$i.err
SDNQ_BENCH_WEIGHT_ARENA=1 timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r5/d1_arena$i.json 2> gpurun_out/r5/d1_arena$i.err
done
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --activation-pool 8 > gpurun_out/r5/d1_pool.json 2> gpurun_out/r5/d1_pool.err
SDNQ_BENCH_WEIGHT_ARENA=1 timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --activation-pool 8 > gpurun_out/r5/d1_pool_arena.json 2> gpurun_out/r5/d1_pool_arena.err
for f in base1 arena1 base2 arena2 pool pool_arena; do python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r5/d1_$f.json").read().strip().splitlines()[-1]); print("$f", d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"])
except Exception as e: print("$f", "ERR", e)
PY
done
tail -3 gpurun_out/r5/d1_arena1.err
