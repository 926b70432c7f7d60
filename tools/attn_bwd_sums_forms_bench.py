"""attn_bwd_fused64_kernel at 1024 x 16 x 257 with / without the per-item token sums (antmmf_attention_bwd_sums), and -- lab library, ANTMMF_ATTN_SUMS_ABL, timing only -- with
parts of the sums machinery removed: 16 no per-chunk dQ sums, 32 no end-of-item dK / dV partials, 64 no finalisation (48 / 112: combinations).

    [ANTMMF_HIP_LIB=.../libantmmf_hip_lab.so ANTMMF_ATTN_SUMS_ABL=n] python tools/attn_bwd_sums_forms_bench.py
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ant-multi-modal-framework_amd"))
from antmmf.hip import ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    dev = torch.device("cuda:0")
    shapes = [tuple(int(x) for x in sp.split("x")) for sp in os.environ.get("ATTN_BENCH_SHAPES", "1024x16x257").split(",")]
    for B, h, N in shapes:
        bench(dev, B, h, N)


def bench(dev, B, h, N):
    qkv = torch.randn(B, N, 3 * h * 64, device=dev).bfloat16()
    q, k, v = qkv[..., :h * 64], qkv[..., h * 64:2 * h * 64], qkv[..., 2 * h * 64:]
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = dqkv[..., :h * 64], dqkv[..., h * 64:2 * h * 64], dqkv[..., 2 * h * 64:]
    o, lse = ops.attention_fwd(q, k, v, h, 0.125)
    do = torch.randn_like(o)
    sums = torch.empty(B, 3 * h * 64, device=dev)
    abl = os.environ.get("ANTMMF_ATTN_SUMS_ABL", "0")
    for rep in range(2):
        for name, fn in (("plain", lambda: ops.attention_bwd(q, k, v, o, lse, do, h, 0.125, dq=dq, dk=dk, dv=dv)),
                         ("sums q|k|v", lambda: ops.attention_bwd(q, k, v, o, lse, do, h, 0.125, dq=dq, dk=dk, dv=dv, sums=sums)),
                         ("sums q|k", lambda: ops.attention_bwd(q, k, v, o, lse, do, h, 0.125, dq=dq, dk=dk, dv=dv, sums=sums, sums_v=False))):
            print(json.dumps({"shape": f"{B}x{h}x{N}", "abl": abl, "rep": rep, "form": name, "ms": round(timeit(fn), 4)}), flush=True)


if __name__ == "__main__":
    main()
