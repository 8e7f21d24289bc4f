python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_gpu_r2.log 2>&1; tail -12 gpurun_out/pytest_gpu_r2.log
python tools/bench_float.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_float.log
