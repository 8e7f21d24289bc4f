python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline 2>gpurun_out/dist_err.log | cut -c1-330; tail -2 gpurun_out/dist_err.log
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --tp 2>gpurun_out/dist_err2.log | cut -c1-330; tail -2 gpurun_out/dist_err2.log
python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_gpu_r2.log 2>&1; tail -4 gpurun_out/pytest_gpu_r2.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for w in sdxl_fp8 linear_int8 flux_int8_svd; do python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tee gpurun_out/bench_$w.json | cut -c1-230; done
