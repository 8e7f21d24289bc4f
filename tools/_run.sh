mkdir -p gpurun_out/r5
bash tools/prof_bench.sh r5/prof_compile --launch compile --steps 20 --warmup 3 2>&1 | tail -22
echo ---- graph
bash tools/prof_bench.sh r5/prof_graph --steps 20 --warmup 3 2>&1 | tail -22
