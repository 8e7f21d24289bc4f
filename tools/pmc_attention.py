#!/usr/bin/env python3
"""Minimal target for rocprofv3 (kernel trace or --pmc passes): a few quantized-attention calls at the SDXL / FLUX sizes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdnq_amd import attention as A  # noqa: E402

dev = torch.device("cuda:0")
for (h, n, d) in ((10, 4096, 64), (24, 4608, 128)):
    q, k, v = (torch.randn(1, h, n, d, device=dev, dtype=torch.bfloat16) for _ in range(3))
    for _ in range(3):
        A.sdnq_hip_atten(q, k, v)
torch.cuda.synchronize()
