"""Row-sharded loss kernels at BASELINE config 2's size (rank-local slabs of the global batch: 1024 rows x 8192 columns, fp32): achieved GB/s against the HBM roofline.
Algorithmic bytes: forward reads the slab(s) once; backward reads them once and writes the gradient slab(s) once.    python tools/loss_bench.py [rows=1024] [cols=8192]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ant-multi-modal-framework_amd"))
from antmmf.hip import ops  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
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
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cuda").manual_seed(0)
    x = (torch.rand(R, W, generator=g, device=dev) * 2 - 1) * 0.3
    x2 = (torch.rand(R, W, generator=g, device=dev) * 2 - 1) * 0.3
    ls = torch.tensor([2.6593], device=dev)
    coef = torch.full((R,), 1.0 / W, device=dev)
    rows, lse = ops.softmax_ce_fwd(x, 0, ls)
    dscale = torch.zeros(1, device=dev)
    lr, denom = ops.milnce_fwd(x, x2, 1, 0)
    out = []
    slab = R * W * 4
    for name, fn, nbytes in (
        ("softmax_ce_fwd", lambda: ops.softmax_ce_fwd(x, 0, ls), slab),
        ("softmax_ce_bwd", lambda: ops.softmax_ce_bwd(x, lse, coef, 0, ls, dscale=dscale, out_dtype=torch.float32), 2 * slab),
        ("milnce_fwd", lambda: ops.milnce_fwd(x, x2, 1, 0), 2 * slab),
        ("milnce_bwd", lambda: ops.milnce_bwd(x, x2, denom, coef, 1, 0, out_dtype=torch.float32), 4 * slab),
    ):
        ms = timeit(fn)
        d = {"kernel": name, "rows": R, "cols": W, "ms": round(ms, 4), "algorithmic_mb": round(nbytes / 1e6, 1), "gbs": round(nbytes / ms / 1e6, 1),
             "frac_of_8tbs": round(nbytes / ms / 1e6 / 8000, 3)}
        print(json.dumps(d), flush=True)
        out.append(d)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/loss_bench.jsonl", "w") as f:
        for d in out:
            f.write(json.dumps(d) + "\n")


if __name__ == "__main__":
    main()
