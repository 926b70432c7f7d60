"""LayerNorm kernel precision against an fp64 reference (debugging aid): python tools/ln_precision.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ant-multi-modal-framework_amd"))
from antmmf.hip import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for dtype in (torch.float32, torch.bfloat16):
    for cols in (128, 768, 1024):
        rows = 512
        x = (torch.randn(rows, cols) * 2 + 0.5).to(dtype)
        dy = torch.randn(rows, cols).to(dtype)
        g = 1 + 0.1 * torch.randn(cols); b = 0.1 * torch.randn(cols)
        xd = x.double().requires_grad_(True); gd = g.double().requires_grad_(True); bd = b.double().requires_grad_(True)
        yd = torch.nn.functional.layer_norm(xd, (cols,), gd, bd, 1e-5)
        yd.backward(dy.double())
        y, mean, rstd = ops.layernorm_fwd(x.to(dev), g.to(dev), b.to(dev), 1e-5)
        dg = torch.zeros(cols, device=dev); db = torch.zeros(cols, device=dev)
        dx = ops.layernorm_bwd(dy.to(dev), x.to(dev), mean, rstd, g.to(dev), dg, db)
        def err(a, ref):
            a = a.double().cpu(); return float((a - ref).abs().max() / ref.abs().max()), float((a - ref).norm() / ref.norm())
        print(str(dtype)[6:], cols, "y", err(y, yd.detach()), "dx", err(dx, xd.grad), "dgamma", err(dg, gd.grad), "dbeta", err(db, bd.grad), "mean", err(mean, x.double().mean(-1)),
              "rstd", err(rstd, 1 / torch.sqrt(x.double().var(-1, unbiased=False) + 1e-5)))
