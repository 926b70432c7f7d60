#!/usr/bin/env python
"""Per-kernel micro-benchmarks on the MI355X (HIP events on torch's current stream): achieved TFLOP/s or GB/s
against the roofline that bounds each kernel.  Writes one JSON line per kernel to stdout."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ant-multi-modal-framework_amd"))
from antmmf.hip import ops  # noqa: E402

DEV = torch.device("cuda:0")
BF = torch.bfloat16
PEAK_TF, PEAK_GBS = 2500.0, 8000.0


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def report(name, secs, flops=None, bytes_=None, **kw):
    d = dict(kernel=name, ms=round(secs * 1e3, 4), **kw)
    if flops:
        d["tflops"] = round(flops / secs / 1e12, 1)
        d["frac_mfma"] = round(flops / secs / 1e12 / PEAK_TF, 4)
    if bytes_:
        d["gbs"] = round(bytes_ / secs / 1e9, 1)
        d["frac_hbm"] = round(bytes_ / secs / 1e9 / PEAK_GBS, 4)
    print(json.dumps(d), flush=True)


def main():
    tokens = 257 * 256  # 65.8k tokens (B = 256 images of ViT-L/14)
    d = 1024
    X = torch.randn(tokens, d, device=DEV).to(BF)
    for (n, k, tag) in ((4 * d, d, "fc1"), (d, 4 * d, "fc2"), (3 * d, d, "qkv"), (d, d, "out")):
        A = torch.randn(tokens, k, device=DEV).to(BF)
        W = (torch.randn(n, k, device=DEV) * k ** -0.5).to(BF)
        bias = torch.randn(n, device=DEV)
        out = torch.empty(tokens, n, dtype=BF, device=DEV)
        report(f"gemm.fwd.{tag}", timeit(lambda: ops.gemm(A, W, out=out, bias=bias)), flops=2.0 * tokens * n * k, M=tokens, N=n, K=k)
        report(f"torch.matmul.{tag}", timeit(lambda: torch.matmul(A, W.t(), out=out)), flops=2.0 * tokens * n * k)
        dY = torch.randn(tokens, n, device=DEV).to(BF)
        dX = torch.empty(tokens, k, dtype=BF, device=DEV)
        report(f"gemm.dgrad.{tag}", timeit(lambda: ops.gemm(dY, W, out=dX, q_rmajor=True)), flops=2.0 * tokens * n * k)
        dW = torch.zeros(n, k, device=DEV)
        sk = 1 if (n // 128) * (k // 128) >= 256 else 4
        report(f"gemm.wgrad.{tag}", timeit(lambda: ops.gemm_wgrad_(dW, dY, A, sk)), flops=2.0 * tokens * n * k)
        Wt_ = W.t().contiguous()
        report(f"gemm.dgrad_pretransposed.{tag}", timeit(lambda: ops.gemm(dY, Wt_, out=dX)), flops=2.0 * tokens * n * k)
        report(f"torch.wgrad.{tag}", timeit(lambda: torch.matmul(dY.t(), A)), flops=2.0 * tokens * n * k)
    g, b = torch.ones(d, device=DEV), torch.zeros(d, device=DEV)
    y, mean, rstd = ops.layernorm_fwd(X, g, b, 1e-5)
    report("layernorm.fwd", timeit(lambda: ops.layernorm_fwd(X, g, b, 1e-5)), bytes_=2.0 * X.numel() * 2)
    dg, db = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    report("layernorm.bwd", timeit(lambda: ops.layernorm_bwd(y, X, mean, rstd, g, dg, db)), bytes_=3.0 * X.numel() * 2)
    U = torch.randn(tokens, 4 * d, device=DEV).to(BF)
    report("act.gelu.fwd", timeit(lambda: ops.act_fwd(U, "gelu")), bytes_=2.0 * U.numel() * 2)
    for (B, h, N) in ((256, 16, 257), (256, 16, 77), (256, 12, 197)):
        qkv = torch.randn(B, N, 3 * h * 64, device=DEV).to(BF)
        q, k, v = qkv[..., :h * 64], qkv[..., h * 64:2 * h * 64], qkv[..., 2 * h * 64:]
        o, lse = ops.attention_fwd(q, k, v, h, 0.125)
        fl = 4.0 * B * h * N * N * 64
        report(f"attention.fwd.N{N}", timeit(lambda: ops.attention_fwd(q, k, v, h, 0.125)), flops=fl, B=B, heads=h)
        do = torch.randn_like(o)
        report(f"attention.bwd.N{N}", timeit(lambda: ops.attention_bwd(q, k, v, o, lse, do, h, 0.125)), flops=2.5 * fl)
        qh, kh, vh = (t.reshape(B, N, h, 64).transpose(1, 2) for t in (q, k, v))
        report(f"torch.sdpa.fwd.N{N}", timeit(lambda: torch.nn.functional.scaled_dot_product_attention(qh, kh, vh)), flops=fl)
    Bl, Bg = 1024, 8192
    Rm = torch.randn(Bl, Bg, device=DEV) * 0.1
    Cm = torch.randn(Bl, Bg, device=DEV) * 0.1
    lr, den = ops.milnce_fwd(Rm, Cm, 1, 0)
    report("milnce.fwd", timeit(lambda: ops.milnce_fwd(Rm, Cm, 1, 0)), bytes_=2.0 * Rm.numel() * 4)
    coef = torch.full((Bl,), 1.0 / Bg, device=DEV)
    report("milnce.bwd", timeit(lambda: ops.milnce_bwd(Rm, Cm, den, coef, 1, 0)), bytes_=2.0 * Rm.numel() * 6)


if __name__ == "__main__":
    main()
