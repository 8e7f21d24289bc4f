for p in 0 4 0 4; do
  timeout 900 python bench.py --no-cpu-baseline --activation-pool $p 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pool=$p', d['ms_per_step'], d['config'].get('eager_ms_per_step'), d['config'].get('activation_buffers'))"
done
