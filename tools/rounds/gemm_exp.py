#!/usr/bin/env python
"""GEMM experiments: effect of the operand row stride (power-of-two vs padded) on the LDS-DMA kernels."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ant-multi-modal-framework_amd"))
from antmmf.hip import ops  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16


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


tokens = 257 * 256
for (n, k, tag) in ((4096, 1024, "fc1"), (1024, 4096, "fc2"), (1024, 1024, "out")):
    for pad in (0, 64, 8):
        Xb = torch.randn(tokens, k + pad, device=dev).to(BF)
        Wb = (torch.randn(n, k + pad, device=dev) * k ** -0.5).to(BF)
        X, W = Xb[:, :k], Wb[:, :k]
        for opad in (0, 64):
            outb = torch.empty(tokens, n + opad, dtype=BF, device=dev)
            out = outb[:, :n]
            t = timeit(lambda: ops.gemm(X, W, out=out))
            print(json.dumps(dict(kernel=f"nt.{tag}", ld_pad=pad, out_pad=opad, ms=round(t * 1e3, 4), tflops=round(2.0 * tokens * n * k / t / 1e12, 1))), flush=True)
    # wgrad with padded strides
    for pad in (0, 64):
        dYb = torch.randn(tokens, n + pad, device=dev).to(BF)
        Xb = torch.randn(tokens, k + pad, device=dev).to(BF)
        dW = torch.zeros(n, k, device=dev)
        t = timeit(lambda: ops.gemm(dYb[:, :n], Xb[:, :k], out=dW, p_rmajor=True, q_rmajor=True, accumulate=True))
        print(json.dumps(dict(kernel=f"tn.{tag}", ld_pad=pad, ms=round(t * 1e3, 4), tflops=round(2.0 * tokens * n * k / t / 1e12, 1))), flush=True)
