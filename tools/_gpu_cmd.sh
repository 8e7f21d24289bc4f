python bench.py --no-cpu-baseline --fuse-projections 2>/dev/null | tee gpurun_out/bench_sdxl_fused.json | cut -c1-330
python bench.py --workload flux_int4_had --steps 5 --warmup 2 --no-cpu-baseline --fuse-projections 2>/dev/null | tee gpurun_out/bench_flux_int4_fused.json | cut -c1-330
python bench.py --workload flux_int4_had --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tee gpurun_out/bench_flux_int4.json | cut -c1-330
