python -m pytest tests -m gpu -q --tb=short -k "conv or im2col" > gpurun_out/pytest_conv.log 2>&1; tail -8 gpurun_out/pytest_conv.log
python bench.py --workload sdxl_conv_int8 --steps 10 --warmup 2 2>gpurun_out/bench_err.log | tee gpurun_out/bench_conv_b.json | cut -c1-200; tail -3 gpurun_out/bench_err.log
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c -o conv -- python $R/bench.py --workload sdxl_conv_int8 --steps 5 --warmup 1 > /dev/null 2>&1
head -8 /tmp/prof_c/conv_kernel_stats.csv | cut -c1-100,180-330; cp /tmp/prof_c/conv_kernel_stats.csv $R/gpurun_out/r01_bench_sdxl_conv_kernel_stats.csv
