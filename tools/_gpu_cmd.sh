python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_gpu_r2.log 2>&1; tail -25 gpurun_out/pytest_gpu_r2.log
