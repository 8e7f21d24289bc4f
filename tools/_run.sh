for k in 20 200 2000; do
  timeout 900 python bench.py --steps $k --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps', d['steps'], 'ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'])"
done
