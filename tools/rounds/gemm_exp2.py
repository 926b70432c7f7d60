#!/usr/bin/env python
"""GEMM experiment: operand data content (DVFS) -- randn vs constant vs small-positive, own kernel vs hipBLASLt."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ant-multi-modal-framework_amd"))
from antmmf.hip import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
tokens, n, k = 257 * 256, 4096, 1024
fl = 2.0 * tokens * n * k
for name, mk in (("randn", lambda s: torch.randn(s, device=dev)), ("zeros", lambda s: torch.zeros(s, device=dev)),
                 ("const", lambda s: torch.full(s, 0.01, device=dev)), ("uniform_pos", lambda s: torch.rand(s, device=dev) * 0.0078 + 0.0078)):
    X = mk((tokens, k)).to(BF); W = mk((n, k)).to(BF); out = torch.empty(tokens, n, dtype=BF, device=dev)
    t1 = timeit(lambda: ops.gemm(X, W, out=out))
    t2 = timeit(lambda: torch.matmul(X, W.t(), out=out))
    print(json.dumps(dict(data=name, own_tflops=round(fl / t1 / 1e12, 1), hipblaslt_tflops=round(fl / t2 / 1e12, 1))), flush=True)
