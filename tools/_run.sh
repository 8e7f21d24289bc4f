mkdir -p gpurun_out/r6
timeout 1200 python -m pytest tests/test_attention.py -x -q -m gpu -k variants 2>&1 | grep -v "^  File\|Extension modules" | tail -12 | tee gpurun_out/r6/pytest_attn_var.txt
timeout 900 python tools/attn_variants_lab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/attn_variants_lab.txt
