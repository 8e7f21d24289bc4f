mkdir -p gpurun_out/r6
timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/r6/bench_default_final.json
python -c "
import json; d=json.load(open('gpurun_out/r6/bench_default_final.json')); r=d['roofline']
print(d['ms_per_step'], d['config'].get('eager_ms_per_step'), r['frac'], r['traffic_stale'], d['cpu_baseline']['value'])"
