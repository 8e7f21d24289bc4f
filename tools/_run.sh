for i in 1 2; do
timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('eager_ms_per_step'), d['config'].get('eager_fast_path'), d['config'].get('eager_error'))"
timeout 900 python bench.py --no-cpu-baseline --launch eager 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('launch=eager', d['ms_per_step'])"
done
