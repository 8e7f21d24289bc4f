mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_fastpath.py tests/test_capture.py -x -q 2>&1 | grep -v "^  File\|Extension modules" | tail -25
