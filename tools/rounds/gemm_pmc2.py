#!/usr/bin/env python
"""Launches ONE plain NT GEMM shape (I = 1024 x 257, J, R from argv) a few times for rocprofv3 --pmc passes; the kernel variant comes from ANTMMF_GEMM_VARIANT."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ant-multi-modal-framework_amd"))
from antmmf.hip import ops  # noqa: E402

dev = torch.device("cuda:0")
J, R, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
tokens = 257 * 1024
X = torch.randn(tokens, R, device=dev).to(torch.bfloat16)
W = (torch.randn(J, R, device=dev) * R ** -0.5).to(torch.bfloat16)
U = torch.empty(tokens, J, dtype=torch.bfloat16, device=dev)
for _ in range(n):
    ops.gemm(X, W, out=U)
torch.cuda.synchronize()
print("done")
