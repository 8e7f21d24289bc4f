import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from sdnq_amd import attention as A
from oracle import oracle as O
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for dt, tag in ((torch.bfloat16, "bf16"), (torch.float16, "f16")):
    for d in (64, 128, 40):
        for kn in (320, 333, 352, 77, 64, 96, 1024, 1000):
            q, k, v = (torch.randn(1, 2, n, d, generator=g).to(dt) for n in (200, kn, kn))
            out = A.sdnq_hip_atten(q.to(dev), k.to(dev), v.to(dev)).float().cpu().numpy()
            ref = O.attention(q.float().numpy(), k.float().numpy(), v.float().numpy(), tag)
            exact = torch.nn.functional.scaled_dot_product_attention(q.double(), k.double(), v.double()).numpy()
            sc = np.abs(exact).max()
            print(tag, "d", d, "kn", kn, "hip-vs-oracle %.4f  hip-vs-exact %.4f  oracle-vs-exact %.4f" % (np.abs(out - ref).max() / sc, np.abs(out - exact).max() / sc, np.abs(ref - exact).max() / sc))
