for v in 0 1 0 1; do
  SDNQ_HIP_FUSED_ROWQUANT_FP8=$v timeout 900 python bench.py --workload sdxl_fp8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('FP8 fused=$v', d['ms_per_step'], d['config'].get('eager_ms_per_step'), d['config'].get('one_launch_linears'))"
done
