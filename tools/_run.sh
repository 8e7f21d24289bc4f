mkdir -p gpurun_out/r6
{
for seed in 14 15; do echo "== fuzz_host_state seed $seed"; timeout 600 python tools/fuzz_host_state.py $seed 2000 2>&1 | tail -2; done
for seed in 23 24 25; do echo "== fuzz_modes (with the float16 matmul) seed $seed"; timeout 900 python tools/fuzz_modes.py $seed 200 2>&1 | tail -4; done
} | grep -v amdgpu.ids | tee gpurun_out/r6/fuzz_round6b.txt
