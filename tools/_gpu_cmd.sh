python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_gpu_r2.log 2>&1; tail -8 gpurun_out/pytest_gpu_r2.log
python bench.py --workload flux_int4_had --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_flux_int4.json 2>gpurun_out/bench_err.log; cat gpurun_out/bench_flux_int4.json
