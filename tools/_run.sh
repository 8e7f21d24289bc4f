mkdir -p gpurun_out/r5
timeout 600 python tools/host_cost.py > gpurun_out/r5/h1_host_cost.txt 2>&1; cat gpurun_out/r5/h1_host_cost.txt
timeout 600 python tools/eager_call_cost.py 2>&1 | head -45 > gpurun_out/r5/h1_eager_call.txt; cat gpurun_out/r5/h1_eager_call.txt
