"""Attention micro-bench at the shapes of the ViT-L/14 step (B = 1024, 16 heads x 64; 257 image / 77 text tokens, packed qkv rows).

    python tools/attn_bench.py [tag] [iters]

One JSON line per kernel pair: ms, TFLOP/s (4 N^2 64 per (b, h) forward, 2.5x backward), algorithmic GB/s.
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ant-multi-modal-framework_amd"))
from antmmf.hip import ops  # noqa: E402


def timeit(fn, iters, warm=2):
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
    tag = sys.argv[1] if len(sys.argv) > 1 else ""
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    out = []
    # ATTN_BENCH_SHAPES="1024x16x257,1024x12x197": batch x heads x tokens per entry (default: the two towers of the ViT-L/14 step)
    shapes = [tuple(int(x) for x in sp.split("x")) for sp in os.environ.get("ATTN_BENCH_SHAPES", "1024x16x257,1024x16x77").split(",")]
    for B, h, N in shapes:
        qkv = torch.randn(B, N, 3 * h * 64, device=dev).to(torch.bfloat16)
        q, k, v = qkv[..., :h * 64], qkv[..., h * 64:2 * h * 64], qkv[..., 2 * h * 64:]
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = dqkv[..., :h * 64], dqkv[..., h * 64:2 * h * 64], dqkv[..., 2 * h * 64:]
        o, lse = ops.attention_fwd(q, k, v, h, 0.125)
        do = torch.randn_like(o)
        sums = torch.empty(B, 3 * h * 64, dtype=torch.float32, device=dev)
        bias_grad = torch.zeros(3 * h * 64, dtype=torch.float32, device=dev)
        fl = 4.0 * B * h * N * N * 64
        e = B * N * h * 64 * 2  # bytes of one [B, N, 1024] bf16 tensor
        for name, fn, flops, nbytes in (
            (f"attention.fwd.N{N}", lambda: ops.attention_fwd(q, k, v, h, 0.125), fl, 4 * e),
            (f"attention.bwd.N{N}", lambda: ops.attention_bwd(q, k, v, o, lse, do, h, 0.125, dq=dq, dk=dk, dv=dv), 2.5 * fl, 8 * e),
            # the same with the per-item token sums of dQ | dK | dV (round 6), and what they replace: a column-sum pass over the [B * N, 3 * D] gradient
            (f"attention.bwd_sums.N{N}", lambda: ops.attention_bwd(q, k, v, o, lse, do, h, 0.125, dq=dq, dk=dk, dv=dv, sums=sums), 2.5 * fl, 8 * e),
            (f"colsum.dqkv.N{N}", lambda: ops.colsum_(bias_grad, dqkv.view(B * N, 3 * h * 64)), 0.0, 3 * e),
            (f"colsum.sums.N{N}", lambda: ops.colsum_(bias_grad, sums), 0.0, B * 3 * h * 64 * 4),
        ):
            if "sums" in name and not ops.attention_bwd_sums_ok(64, N, N):
                continue
            ms = timeit(fn, iters)
            d = {"tag": tag, "kernel": name, "ms": round(ms, 4), "tflops": round(flops / ms / 1e9, 1), "gbs_min": round(nbytes / ms / 1e6, 1)}
            print(json.dumps(d), flush=True)
            out.append(d)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/attn_bench_{tag or 'run'}.jsonl", "w") as f:
        for d in out:
            f.write(json.dumps(d) + "\n")


if __name__ == "__main__":
    main()
