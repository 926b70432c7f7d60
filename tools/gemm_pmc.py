#!/usr/bin/env python
"""Launches the large GEMMs of one ViT-L/14 layer a few times (for rocprofv3 --pmc passes)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ant-multi-modal-framework_amd"))
from antmmf.hip import ops  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
tokens, d = 257 * 256, 1024
X = torch.randn(tokens, d, device=dev).to(BF)
W1 = (torch.randn(4 * d, d, device=dev) * d ** -0.5).to(BF)
U = torch.empty(tokens, 4 * d, dtype=BF, device=dev)
dY = torch.randn(tokens, 4 * d, device=dev).to(BF)
dW = torch.zeros(4 * d, d, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    ops.gemm(X, W1, out=U)                                                           # fc1 forward  (NT, 256^2 DMA)
    ops.gemm(dY, X, out=dW, p_rmajor=True, q_rmajor=True, accumulate=True)           # fc1 wgrad    (TN, DMA + tr read)
torch.cuda.synchronize()
print("done")
