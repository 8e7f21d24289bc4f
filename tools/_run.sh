mkdir -p gpurun_out/r5
(
for s in 51 52; do timeout 900 python tools/fuzz_linear.py $s 150 2>&1 | tail -1; done
for s in 53 54; do timeout 900 python tools/fuzz_ops.py $s 150 2>&1 | tail -1; done
timeout 1200 python tools/fuzz_tiles.py 55 40 2>&1 | tail -1
for s in 56 57; do timeout 900 python tools/fuzz_host_state.py $s 600 2>&1 | tail -1; done
timeout 900 python tools/fuzz_attention_routes.py 58 60 2>&1 | tail -1
for s in 59 60; do timeout 900 python tools/fuzz_w8a16.py $s 100 2>&1 | tail -1; done
for s in 61 62; do timeout 900 python tools/fuzz_fused.py $s 300 2>&1 | tail -1; done
) 2>&1 | grep -v amdgpu | cut -c1-200 > gpurun_out/r5/n1_fuzz_all.txt
cat gpurun_out/r5/n1_fuzz_all.txt
