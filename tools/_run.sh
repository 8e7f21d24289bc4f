mkdir -p gpurun_out/r6
timeout 1200 python -m pytest tests/test_attention.py -x -q -m gpu 2>&1 | grep -v "^  File\|Extension modules" | tail -30 | tee gpurun_out/r6/pytest_attn_var.txt
