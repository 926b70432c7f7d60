"""Per-op timing of the M2 feed-forward at the bench size, separate LayerNorm kernels vs the sub-LN fold (same process, same buffers, interleaved rounds).
usage (GPU box): python tools/ffn_fold_bench.py [pairs=1024] [tokens_per_pair=257] [rounds=5]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ant-multi-modal-framework_amd"))
from antmmf.hip import ops  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
tpp = int(sys.argv[2]) if len(sys.argv) > 2 else 257
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
T, d, ff = pairs * tpp, 1024, 4096
dev = torch.device("cuda:0")
BF = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, scale=1.0: torch.randn(*s, device=dev, generator=g) * scale
x, res, dy = rn(T, d).to(BF), rn(T, d).to(BF), rn(T, d).to(BF)
W1, b1 = rn(ff, d, scale=d ** -0.5), rn(ff, scale=0.1)
W2, b2 = rn(d, ff, scale=ff ** -0.5), rn(d, scale=0.1)
gam, bet = 1 + 0.1 * rn(ff), 0.1 * rn(ff)
W1b, W2b, W2tb = W1.to(BF), W2.to(BF), W2.to(BF).t().contiguous()


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


res_ms = {}
# ---- separate kernels
u = ops.gemm(x, W1b, bias=b1)
gn, mf, rf = ops.layernorm_fwd(u, gam, bet, 1e-5, act="gelu")
y = ops.gemm(gn, W2b, bias=b2, residual=res)
dgn = ops.gemm(dy, W2tb)
dgw, dgb, dxs = torch.zeros(ff, device=dev), torch.zeros(ff, device=dev), torch.zeros(ff, device=dev)
dW2 = torch.zeros(d, ff, device=dev)
# ---- fold
w2g, c, b2f = ops.ffn_prepare_w2(W2, gam, bet, b2)
w2gt = ops.transpose_bf16(w2g)
z, dact, st = ops.ffn_fc1_fwd(x, W1b, b1, "gelu", 1e-5)
yf = ops.ffn_fc2_fwd(z, w2g, c, b2f, st, res)
s_col, cs_col = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
rowv4, dys = ops.ffn_bwd_rows(dy, yf, res, b2f, c, st, ff, s_col, cs_col)
db1 = torch.zeros(ff, device=dev)
Gm = torch.zeros(d, ff, device=dev)
print(json.dumps({"check": "fold y vs separate y", "max_abs_diff": float((yf.float() - y.float()).abs().max()), "rms_y": float(y.float().pow(2).mean().sqrt())}), flush=True)
cases = {
    "sep.fc1 (bias)": lambda: ops.gemm(x, W1b, bias=b1, out=u),
    "sep.ln_fwd (gelu + LN)": lambda: ops.layernorm_fwd(u, gam, bet, 1e-5, act="gelu"),
    "sep.fc2 (bias + residual)": lambda: ops.gemm(gn, W2b, bias=b2, residual=res, out=y),
    "sep.dgrad fc2 (plain)": lambda: ops.gemm(dy, W2tb, out=dgn),
    "sep.ln_bwd (LN + gelu backward, db1)": lambda: ops.layernorm_bwd(dgn, u, mf, rf, gam, dgw, dgb, act="gelu", dxsum=dxs),
    "sep.wgrad fc2": lambda: ops.gemm_wgrad_(dW2, dy, gn),
    "fold.fc1 (gelu, gelu', row sums) + stats": lambda: ops.ffn_fc1_fwd(x, W1b, b1, "gelu", 1e-5),
    "fold.fc2 (row affine + residual)": lambda: ops.ffn_fc2_fwd(z, w2g, c, b2f, st, res),
    "fold.bwd rows (d-wide)": lambda: ops.ffn_bwd_rows(dy, yf, res, b2f, c, st, ff, s_col, cs_col),
    "fold.dgrad fc2 (LN + gelu backward, db1 partials)": lambda: ops.ffn_fc2_dgrad(dy, w2gt, z, dact, rowv4, db1),
    "fold.wgrad fc2 (Gm) + post": lambda: (Gm.zero_(), ops.gemm_wgrad_(Gm, dys, z), ops.ffn_wgrad_post_(dW2, Gm, W2, gam, bet, s_col, cs_col, dgw, dgb)),
    "fold.prepare W2 + transpose (per step)": lambda: ops.transpose_bf16(ops.ffn_prepare_w2(W2, gam, bet, b2)[0]),
}
acc = {k: [] for k in cases}
for _ in range(rounds):
    for k, fn in cases.items():
        acc[k].append(timed(fn))
for k, v in acc.items():
    v.sort()
    print(json.dumps({"op": k, "tokens": T, "ms_med": round(v[len(v) // 2], 4), "ms_min": round(v[0], 4)}), flush=True)
med = {k: sorted(v)[len(v) // 2] for k, v in acc.items()}
sep = sum(v for k, v in med.items() if k.startswith("sep."))
fold = sum(v for k, v in med.items() if k.startswith("fold.") and "per step" not in k)
print(json.dumps({"summary": "feed-forward fwd + bwd pieces that differ", "tokens": T, "separate_ms": round(sep, 3), "fold_ms": round(fold, 3), "saved_ms": round(sep - fold, 3)}))
