import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from tests.golden_util import Case
from tests.modules_util import module_from_case, to_f32_numpy
from sdnq_amd import ops, linear
dev = torch.device("cuda:0")
for name in ["int4_had256_qmm_bf16", "int8_had256_qmm_bf16", "int8_had64_k192_qmm_bf16"]:
    c = Case(name)
    mod = module_from_case(c, dev)
    for M in c.ms():
        if M >= 32: continue
        x = c.torch_tensor(f"x_{M}", device=dev)
        ref = c.f32(f"y_{M}")
        ya = to_f32_numpy(mod(x))
        st = linear._state(mod)
        dq = mod.sdnq_dequantizer
        xr = ops.hadamard(x.reshape(-1, dq.in_features), dq.hadamard_group_size)
        yb = to_f32_numpy(ops.linear_skinny(st.qw, xr, mod.bias, 0))
        sc = np.abs(ref).max()
        print(name, M, "unrotate-W: max", np.abs(ya - ref).max() / sc, "l2", np.linalg.norm(ya - ref) / np.linalg.norm(ref),
              " rotate-x: max", np.abs(yb.reshape(ref.shape) - ref).max() / sc, "l2", np.linalg.norm(yb.reshape(ref.shape) - ref) / np.linalg.norm(ref))
