#!/usr/bin/env python
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
X = (torch.rand(tokens, k, device=dev) * 0.0078 + 0.0078).to(BF); W = (torch.rand(n, k, device=dev) * 0.0078 + 0.0078).to(BF)
out = torch.empty(tokens, n, dtype=BF, device=dev); outf = torch.empty(tokens, n, dtype=torch.float32, device=dev)
bias = torch.randn(n, device=dev)
res = torch.randn(tokens, n, device=dev).to(BF)
for name, fn in (("plain bf16 out", lambda: ops.gemm(X, W, out=out)), ("bias", lambda: ops.gemm(X, W, out=out, bias=bias)),
                 ("bias+gelu", lambda: ops.gemm(X, W, out=out, bias=bias, act="gelu")), ("bias+residual", lambda: ops.gemm(X, W, out=out, bias=bias, residual=res)),
                 ("fp32 out", lambda: ops.gemm(X, W, out=outf)), ("hipblaslt", lambda: torch.matmul(X, W.t(), out=out))):
    t = timeit(fn)
    print(json.dumps(dict(case=name, ms=round(t * 1e3, 4), tflops=round(fl / t / 1e12, 1))), flush=True)
