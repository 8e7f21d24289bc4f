mkdir -p gpurun_out/r6
{
for seed in 101 102 103 104; do echo "== fuzz_modes seed $seed"; timeout 900 python tools/fuzz_modes.py $seed 300 2>&1 | tail -3; done
for seed in 111 112 113; do echo "== fuzz_host_state seed $seed"; timeout 600 python tools/fuzz_host_state.py $seed 3000 2>&1 | tail -2; done
echo "== fuzz_conv"; timeout 900 python tools/fuzz_conv.py 121 150 2>&1 | tail -3
echo "== fuzz_w8a16"; timeout 900 python tools/fuzz_w8a16.py 131 200 2>&1 | tail -3
echo "== fuzz_tiles"; timeout 1200 python tools/fuzz_tiles.py 141 2>&1 | tail -3
echo "== fuzz_attention_routes"; timeout 900 python tools/fuzz_attention_routes.py 151 2>&1 | tail -3
echo "== fuzz_ops"; timeout 900 python tools/fuzz_ops.py 161 200 2>&1 | tail -3
echo "== fuzz_fused"; timeout 900 python tools/fuzz_fused.py 171 400 2>&1 | tail -2
} | grep -v amdgpu.ids | tee gpurun_out/r6/fuzz_round6c.txt
