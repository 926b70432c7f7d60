#!/usr/bin/env python
"""Launches the step's GEMM shapes on the PRODUCT library a few times for rocprofv3 --pmc passes (tools/gpu_r5i.sh): NT forward shapes with their epilogues and one wgrad."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ant-multi-modal-framework_amd"))
from antmmf.hip import ops  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
tokens = 257 * 1024
g = torch.Generator(device="cuda").manual_seed(1)
for J, R, bias, res in ((4096, 1024, True, False), (1024, 4096, True, True), (1024, 1024, True, True), (3072, 1024, True, False), (1024, 4096, False, False)):
    X = torch.randn(tokens, R, device=dev, generator=g).to(torch.bfloat16)
    W = (torch.randn(J, R, device=dev, generator=g) * R ** -0.5).to(torch.bfloat16)
    b = torch.randn(J, device=dev, generator=g) if bias else None
    r = torch.randn(tokens, J, device=dev, generator=g).to(torch.bfloat16) if res else None
    U = torch.empty(tokens, J, dtype=torch.bfloat16, device=dev)
    for _ in range(n):
        ops.gemm(X, W, out=U, bias=b, residual=r)
    del X, W, U, r
dY = torch.randn(tokens, 4096, device=dev, generator=g).to(torch.bfloat16)
Xa = torch.randn(tokens, 1024, device=dev, generator=g).to(torch.bfloat16)
dW = torch.zeros(4096, 1024, device=dev)
for _ in range(n):
    ops.gemm_wgrad_(dW, dY, Xa)
torch.cuda.synchronize()
print("done")
