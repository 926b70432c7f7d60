#!/usr/bin/env python
"""wgrad (TN ring) timing for the ViT-L/14 shapes; run under different ANTMMF_GEMM_RASTER / ANTMMF_WGRAD_WGS settings."""
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
tokens = 257 * 256
tag = os.environ.get("ANTMMF_GEMM_RASTER", "1") + "/" + os.environ.get("ANTMMF_WGRAD_WGS", "dflt")
for name, n, k in (("fc1", 4096, 1024), ("fc2", 1024, 4096), ("out", 1024, 1024)):
    dY = torch.randn(tokens, n, device=dev).to(BF); X = torch.randn(tokens, k, device=dev).to(BF)
    dW = torch.zeros(n, k, device=dev)
    t = timeit(lambda: ops.gemm_wgrad_(dW, dY, X))
    print(json.dumps(dict(cfg=tag, case=name, ms=round(t * 1e3, 4), tflops=round(2.0 * tokens * n * k / t / 1e12, 1))), flush=True)
