#!/usr/bin/env python
"""Attention kernels vs operand layout at the bench size: packed token-major QKV (row stride 3d), separate token-major tensors (row
stride d), head-major (each (batch, head) slab contiguous: emulated with heads = 1 on [B*H, N, 64] tensors)."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ant-multi-modal-framework_amd"))
from antmmf.hip import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
def timeit(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
B, H, N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 16, 257
d = H * 64
qkv = torch.randn(B, N, 3 * d, device=dev).to(BF)
do = torch.randn(B, N, d, device=dev).to(BF)
lay = {}
lay["packed token-major (ld 3d)"] = (qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], H, do, B)
q, k, v = (qkv[..., i * d:(i + 1) * d].contiguous() for i in range(3))
lay["separate token-major (ld d)"] = (q, k, v, H, do, B)
hm = lambda t: t.view(B, N, H, 64).permute(0, 2, 1, 3).reshape(B * H, N, 64).contiguous()
lay["head-major (contiguous per head)"] = (hm(q), hm(k), hm(v), 1, hm(do), B * H)
for name, (q_, k_, v_, h_, do_, b_) in lay.items():
    o, lse = ops.attention_fwd(q_, k_, v_, h_, 0.125)
    tf = timeit(lambda: ops.attention_fwd(q_, k_, v_, h_, 0.125))
    tb = timeit(lambda: ops.attention_bwd(q_, k_, v_, o, lse, do_, h_, 0.125))
    print(json.dumps(dict(layout=name, B=B, fwd_ms=round(tf, 3), bwd_ms=round(tb, 3))), flush=True)
