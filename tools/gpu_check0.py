#!/usr/bin/env python3
"""First-contact GPU check (development aid, not a test): kernels vs plain torch-CPU restatements
and the committed golden fixtures, plus a first timing of the w8a8 path on SDXL shapes."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdnq_amd import _lib, ops  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
dev = torch.device("cuda:0")
print("device", torch.cuda.get_device_name(0), "supported", _lib.load().sdnq_hip_device_supported(0))


def ref_rowquant_i8(x):
    xf = x.float()
    xs = xf.abs().amax(-1, keepdim=True) / 127
    q = (xf / xs).round().clamp(-128, 127)
    q = torch.nan_to_num(q, nan=0.0)
    return q.to(torch.int8), xs


def check_rowquant():
    for (m, k, dt) in [(100, 640, torch.bfloat16), (33, 1280, torch.float16), (7, 5120, torch.float32), (64, 48, torch.bfloat16)]:
        torch.manual_seed(0)
        x = torch.randn(m, k) * 3
        x[:, 5] *= 20
        x[3] = 0
        x = x.to(dt)
        xq, xs, rs, _ = ops.rowquant(x.to(dev), ops.MM_I8, want_rowsum=True)
        rq, rs_ref = ref_rowquant_i8(x)
        ok = torch.equal(xq.cpu(), rq) and torch.equal(xs.cpu(), rs_ref) and torch.equal(rs.cpu(), rq.sum(-1, dtype=torch.int32))
        print(f"rowquant_i8 M={m} K={k} {dt}: {'OK' if ok else 'MISMATCH'}",
              (xq.cpu() != rq).sum().item(), (xs.cpu() != rs_ref).sum().item())
        xq8, xs8, _, _ = ops.rowquant(x.to(dev), ops.MM_FP8)
        xf = x.float()
        s8 = xf.abs().amax(-1, keepdim=True) / 448
        q8 = torch.nan_to_num(xf / s8).clamp(-448, 448).to(torch.float8_e4m3fn)
        ok = torch.equal(xq8.cpu().view(torch.uint8), q8.view(torch.uint8)) and torch.equal(xs8.cpu(), s8)
        print(f"rowquant_fp8 M={m} K={k} {dt}: {'OK' if ok else 'MISMATCH'}",
              (xq8.cpu().view(torch.uint8) != q8.view(torch.uint8)).sum().item())


def check_gemm_i8():
    for (m, n, k, bias, odt) in [(48, 256, 512, True, torch.bfloat16), (128, 128, 128, False, torch.bfloat16),
                                 (77, 640, 2048, True, torch.bfloat16), (4096, 640, 640, True, torch.bfloat16),
                                 (1000, 336, 144, True, torch.float16), (1024, 1280, 5120, True, torch.float32),
                                 (333, 80, 48, True, torch.bfloat16)]:
        torch.manual_seed(1)
        a = torch.randint(-128, 128, (m, k), dtype=torch.int8)
        b = torch.randint(-128, 128, (n, k), dtype=torch.int8)
        sa = torch.rand(m, 1) * 0.01 + 1e-3
        sb = torch.rand(1, n) * 0.01 + 1e-3
        bv = (torch.randn(n).to(odt) if bias else None)
        acc = torch._int_mm(a, b.t().contiguous()) if (m > 16 and k % 8 == 0 and n % 8 == 0) else (a.int() @ b.int().t())
        if bv is not None:
            ref = torch.addcmul(bv.float(), acc.float() * sa, sb).to(odt)
        else:
            ref = (acc.float() * sa * sb).to(odt)
        out = ops.scaled_mm(ops.MM_I8, a.to(dev), b.to(dev), sa.to(dev), sb.to(dev), None if bv is None else bv.to(dev), odt)
        nbad = (out.cpu().float() != ref.float()).sum().item()
        print(f"gemm_i8 M={m} N={n} K={k} bias={bias} {odt}: {'OK' if nbad == 0 else 'MISMATCH'} bad={nbad}"
              + ("" if nbad == 0 else f" maxdiff={(out.cpu().float() - ref.float()).abs().max().item()}"))


def check_gemm_fp8():
    for (m, n, k) in [(48, 256, 512), (256, 128, 1280), (100, 64, 192)]:
        torch.manual_seed(2)
        a = (torch.randn(m, k) * 50).clamp(-448, 448).to(torch.float8_e4m3fn)
        b = (torch.randn(n, k) * 50).clamp(-448, 448).to(torch.float8_e4m3fn)
        sa = torch.rand(m, 1) * 0.01 + 1e-3
        sb = torch.rand(1, n) * 0.01 + 1e-3
        bv = torch.randn(n).to(torch.bfloat16)
        acc = a.float().double() @ b.float().double().t()
        ref = (acc * sa.double() * sb.double() + bv.double())
        out = ops.scaled_mm(ops.MM_FP8, a.to(dev), b.to(dev), sa.to(dev), sb.to(dev), bv.to(dev), torch.bfloat16)
        err = (out.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
        print(f"gemm_fp8 M={m} N={n} K={k}: rel max err {err:.3e}")


def load_case(name):
    meta = json.load(open(os.path.join(GOLD, f"case_{name}.json")))
    z = np.load(os.path.join(GOLD, f"case_{name}.npz"))

    def get(key):
        info = meta["tensors"].get(key)
        if info is None or info["dtype"] == "none":
            return None
        arr = torch.from_numpy(z[key])
        tag = info["dtype"]
        if tag == "bf16":
            arr = arr.view(torch.bfloat16)
        elif tag == "f16":
            arr = arr.view(torch.float16)
        elif tag == "fp8e4m3":
            arr = arr.view(torch.float8_e4m3fn)
        elif tag == "fp8e5m2":
            arr = arr.view(torch.float8_e5m2)
        st = info.get("stride")
        if st is not None and arr.ndim == 2 and st == [1, arr.shape[0]]:
            arr = arr.t().contiguous().t()
        return arr
    return meta, get


def check_dequant_cases():
    names = [f[5:-5] for f in sorted(os.listdir(GOLD)) if f.startswith("case_") and f.endswith(".json")]
    for name in names:
        meta, get = load_case(name)
        dq = meta["deq"]
        n, k = meta["N"], meta["K"]
        gs = dq["group_size"] if dq["group_size"] > 0 else k
        transposed = dq["use_quantized_matmul"] and not dq["re_quantize_for_matmul"] and not dq["is_packed"]
        w, s, zp, up, down = get("weight"), get("scale"), get("zero_point"), get("svd_up"), get("svd_down")
        mv = lambda t: None if t is None else t.to(dev)
        try:
            qw = ops.make_quant_weight(dq["weights_dtype"], mv(w), mv(s), mv(zp), mv(up), mv(down), n, k, gs, transposed)
            rd = {"bfloat16": torch.bfloat16, "float16": torch.float16, "float32": torch.float32}[dq["result_dtype"]]
            wd = ops.dequant(qw, rd, dq["hadamard_group_size"] if dq["use_hadamard"] else 0)
            ref = get("w_dequant")
            bad = (wd.cpu().float() != ref.float()).sum().item()
            ref32 = get("w_dequant_f32_nohad")
            wd32 = ops.dequant(qw, torch.float32, 0)
            bad32 = (wd32.cpu() != ref32).sum().item()
            msg = f"dequant {name}: result-dtype bad={bad}/{ref.numel()} f32-nohad bad={bad32}"
            if bad:
                msg += f" maxdiff={(wd.cpu().float() - ref.float()).abs().max().item():.3e} refmax={ref.float().abs().max().item():.3e}"
            if get("requant_weight") is not None:
                mm = ops.mm_code(dq["quantized_matmul_dtype"])
                wq, ws = ops.requant(qw, mm)
                rw = get("requant_weight").t().contiguous()
                rs = get("requant_scale").view(-1)
                bw = (wq.cpu().view(torch.uint8) != rw.view(torch.uint8)).sum().item()
                bs = (ws.cpu() != rs).sum().item()
                msg += f" | requant bad_w={bw} bad_s={bs}"
            print(msg)
        except Exception as e:  # noqa: BLE001
            print(f"dequant {name}: EXC {e}")


def check_hadamard():
    z = np.load(os.path.join(GOLD, "hadamard.npz"))
    for dt, tdt in (("bf16", torch.bfloat16), ("f32", torch.float32), ("f16", torch.float16)):
        for n in (64, 128, 256):
            x = torch.from_numpy(z[f"x_{dt}_{n}"])
            y = torch.from_numpy(z[f"y_{dt}_{n}"])
            if dt != "f32":
                x, y = x.view(tdt), y.view(tdt)
            out = ops.hadamard(x.to(dev), n).cpu()
            diff = (out.float() - y.float()).abs().max().item()
            nbad = (out.float() != y.float()).sum().item()
            print(f"hadamard {dt} g={n}: bad={nbad}/{y.numel()} maxdiff={diff:.3e} (ymax {y.float().abs().max().item():.2f})")


def time_sdxl():
    shapes = [(4096, 640, 640), (4096, 5120, 640), (4096, 640, 2560), (1024, 1280, 1280), (1024, 10240, 1280),
              (1024, 1280, 5120), (77, 640, 2048), (77, 1280, 2048), (16384, 8192, 4096)]
    for (m, n, k) in shapes:
        x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
        b = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev)
        sb = torch.rand(n, device=dev) * 0.01
        bias = torch.randn(n, device=dev, dtype=torch.bfloat16)
        xq, xs, _, _ = ops.rowquant(x, ops.MM_I8)
        for _ in range(3):
            ops.scaled_mm(ops.MM_I8, xq, b, xs, sb, bias, torch.bfloat16)
        torch.cuda.synchronize()
        it = 20
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        for _ in range(it):
            xq, xs, _, _ = ops.rowquant(x, ops.MM_I8)
        e1.record()
        for _ in range(it):
            ops.scaled_mm(ops.MM_I8, xq, b, xs, sb, bias, torch.bfloat16)
        e2.record()
        torch.cuda.synchronize()
        tq, tg = e0.elapsed_time(e1) / it * 1e3, e1.elapsed_time(e2) / it * 1e3
        print(f"time M={m} N={n} K={k}: rowquant {tq:.1f} us, gemm {tg:.1f} us -> {2 * m * n * k / tg / 1e6:.1f} TOP/s (eager-launch bound?)")


if __name__ == "__main__":
    t0 = time.time()
    check_rowquant()
    check_gemm_i8()
    check_gemm_fp8()
    check_hadamard()
    check_dequant_cases()
    time_sdxl()
    print("done in", time.time() - t0)
