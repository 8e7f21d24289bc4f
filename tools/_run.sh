mkdir -p gpurun_out/r6
timeout 600 python tools/eager_profile.py sdxl_fp8 2>&1 | grep -v amdgpu.ids | head -45 | tee gpurun_out/r6/eager_profile_fp8.txt
timeout 600 python tools/eager_split.py sdxl_fp8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/eager_split_fp8.txt
