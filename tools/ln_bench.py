"""Row-wise kernel micro-bench at the shapes of the ViT-L/14 step (HIP events through torch; kernels through the C ABI).

    python tools/ln_bench.py [tag]          # ANTMMF_HIP_LIB selects another build of the same ABI for A/B runs

One JSON line per kernel: ms, algorithmic GB/s, fraction of the 8 TB/s HBM peak.
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ant-multi-modal-framework_amd"))
from antmmf.hip import ops  # noqa: E402


def timeit(fn, iters=8, warm=2):
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
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    out = []

    def rep(name, ms, nbytes, **kw):
        d = {"tag": tag, "kernel": name, "ms": round(ms, 4), "gbs": round(nbytes / ms / 1e6, 1), "frac_hbm": round(nbytes / ms / 1e6 / 8000.0, 4)}
        d.update(kw)
        print(json.dumps(d), flush=True)
        out.append(d)

    for label, rows in (("image", 1024 * 257), ("text", 1024 * 77)):
        for cols in (1024, 4096):
            x = torch.randn(rows, cols, device=dev).to(torch.bfloat16)
            dy = torch.randn(rows, cols, device=dev).to(torch.bfloat16)
            g = torch.rand(cols, device=dev) + 0.5
            b = torch.randn(cols, device=dev) * 0.1
            dg, db, dxs = torch.zeros_like(g), torch.zeros_like(g), torch.zeros_like(g)
            e = x.numel() * 2
            act = "gelu" if cols == 4096 else None
            y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-5, act=act)
            rep(f"ln_fwd.{label}.{cols}.act={act}", timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-5, act=act)), 2 * e, rows=rows)
            rep(f"ln_bwd.{label}.{cols}.act={act}.dxsum", timeit(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dg, db, act=act, dxsum=dxs)), 3 * e, rows=rows)
            if cols == 1024:
                rep(f"ln_bwd.{label}.{cols}.plain", timeit(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dg, db)), 3 * e, rows=rows)
                dr = torch.randn(rows, cols, device=dev).to(torch.bfloat16)     # (a tensor of its own: dres = dy would be one stream less)
                rep(f"ln_bwd.{label}.{cols}.dres.dxsum", timeit(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dg, db, dres=dr, dxsum=dxs)), 4 * e, rows=rows)
                rep(f"ln_bwd_renorm.{label}.{cols}.dres.dxsum", timeit(lambda: ops.layernorm_bwd_renorm(dy, x, mean, rstd, g, b, dg, db, dres=dr, dxsum=dxs)), 5 * e, rows=rows)
                del dr
                rep(f"colsum.{label}.{cols}", timeit(lambda: ops.colsum_(dxs, dy)), e, rows=rows)
            else:
                y2, m2, r2 = ops.layernorm_fwd(x, g, b, 1e-5, act=None)
                rep(f"ln_fwd.{label}.{cols}.act=None", timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-5, act=None)), 2 * e, rows=rows)
                rep(f"ln_bwd.{label}.{cols}.act=None.dxsum", timeit(lambda: ops.layernorm_bwd(dy, x, m2, r2, g, dg, db, dxsum=dxs)), 3 * e, rows=rows)
                rep(f"copy.{label}.{cols}", timeit(lambda: y.copy_(x)), 2 * e, rows=rows)
            del x, dy, y
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/ln_bench_{tag or 'run'}.jsonl", "w") as f:
        for d in out:
            f.write(json.dumps(d) + "\n")


if __name__ == "__main__":
    main()
