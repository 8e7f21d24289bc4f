timeout 900 python -m pytest tests/test_torch_ops.py -x -q 2>&1 | tail -2
timeout 900 python -m pytest tests/test_torch_ops.py -x -q -m gpu 2>&1 | tail -2
