"""Backend-agnostic kernel parity cases: each function runs one C-ABI kernel family through
antmmf.hip.ops on `dev` and checks it against the CPU oracle (oracle/, plain torch fp32) on the same
seeded inputs.  Used by tests/test_kernels_emu.py (CPU lane emulator, build container) and
tests/test_kernels_gpu.py (real MI355X, `-m gpu`).

Tolerances: fp32-I/O kernels 1e-5-class; bf16-I/O kernels are compared against the fp32 oracle evaluated
on the SAME bf16-rounded inputs, with rtol 2e-2 and an atol of 2e-2 x the reference tensor's own scale
(bf16 has 8 mantissa bits: 2^-8 = 3.9e-3 per rounding)."""
import math

import torch

from oracle import losses as olosses
from oracle import ops as oops

BF = torch.bfloat16


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def check(name, got, want, rtol, atol_rel):
    got = got.detach().float().cpu()
    want = want.detach().float().cpu()
    assert got.shape == want.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(want.shape)}"
    scale = float(want.abs().max()) if want.numel() else 1.0
    atol = atol_rel * max(scale, 1e-30)
    err = (got - want).abs()
    bad = err > (atol + rtol * want.abs())
    if bad.any() or not torch.isfinite(got).all():
        idx = int(torch.argmax(err.flatten()))
        raise AssertionError(f"{name}: {int(bad.sum())}/{bad.numel()} out of tolerance, max abs err {float(err.max()):.4g} "
                             f"(ref scale {scale:.4g}) at flat index {idx}: got {float(got.flatten()[idx]):.6g} want {float(want.flatten()[idx]):.6g}")
    # The absolute term above scales with the tensor's largest element, so on its own it says little about the SMALL elements.  Second
    # criterion, blind to magnitude: the MEDIAN relative error over the non-zero reference elements -- an indexing / permutation error that only
    # disturbs small entries, or garbage below the absolute tolerance, puts it near 1; rounding noise keeps it near the dtype's epsilon.
    nz = want.abs() > 1e-6 * max(scale, 1e-30)
    if int(nz.sum()) >= 16:
        med = float((err[nz] / want[nz].abs()).median())
        lim = max(4.0 * rtol, 20.0 * atol_rel, 1e-5)
        assert med <= lim, f"{name}: median relative error {med:.4g} > {lim:.4g} over {int(nz.sum())} non-zero elements"


def q(t):  # bf16 quantise, keep fp32 container
    return t.to(BF).float()


# ------------------------------------------------------------------------------ LayerNorm
def case_layernorm(ops, dev, dtype, rows=13, cols=128, eps=1e-5):
    x = rnd((rows, cols), 1, 2.0) + 0.5
    g = 1 + 0.1 * rnd((cols,), 2)
    b = 0.1 * rnd((cols,), 3)
    dy = rnd((rows, cols), 4)
    dres = rnd((rows, cols), 5)
    if dtype == BF:
        x, dy, dres = q(x), q(dy), q(dres)
    xr = x.clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = oops.layer_norm(xr, gr, br, eps)
    yr.backward(dy)
    y, mean, rstd = ops.layernorm_fwd(x.to(dev, dtype), g.to(dev), b.to(dev), eps)
    dgam = torch.zeros(cols, device=dev)
    dbet = torch.zeros(cols, device=dev)
    dxs = torch.zeros(cols, device=dev)
    dx = ops.layernorm_bwd(dy.to(dev, dtype), x.to(dev, dtype), mean, rstd, g.to(dev), dgam, dbet, dres.to(dev, dtype), dxsum=dxs)
    rt, at = (2e-2, 2e-2) if dtype == BF else (1e-5, 1e-5)
    check("ln.dxsum", dxs, (xr.grad + dres).sum(0), 1e-3, (2e-2 if dtype == BF else 1e-4) * max(1.0, rows ** 0.5))
    check("ln.y", y, yr, rt, at)
    check("ln.mean", mean, x.mean(-1), 1e-5, 1e-5)
    check("ln.dx", dx, xr.grad + dres, rt, at)
    check("ln.dgamma", dgam, gr.grad, 1e-4 if dtype != BF else 2e-2, 1e-4 if dtype != BF else 2e-2)
    check("ln.dbeta", dbet, br.grad, 1e-4, 1e-4)
    # backward that re-emits the forward output: same dx / parameter gradients as the plain backward, y identical to layernorm_fwd
    dgam2, dbet2, dxs2 = torch.zeros(cols, device=dev), torch.zeros(cols, device=dev), torch.zeros(cols, device=dev)
    dx2, y2 = ops.layernorm_bwd_renorm(dy.to(dev, dtype), x.to(dev, dtype), mean, rstd, g.to(dev), b.to(dev), dgam2, dbet2, dres.to(dev, dtype), dxsum=dxs2)
    assert torch.equal(y2, y), float((y2.float() - y.float()).abs().max())
    assert torch.equal(dx2, dx), float((dx2.float() - dx.float()).abs().max())
    check("ln.renorm.dgamma", dgam2, dgam, 1e-4, 1e-4)
    check("ln.renorm.dbeta", dbet2, dbet, 1e-4, 1e-4)
    check("ln.renorm.dxsum", dxs2, dxs, 1e-4, 1e-4)
    dx3, y3 = ops.layernorm_bwd_renorm(dy.to(dev, dtype), x.to(dev, dtype), mean, rstd, g.to(dev), b.to(dev))
    assert torch.equal(y3, y)
    check("ln.renorm.dx_nores", dx3, xr.grad, rt, at)


def case_act_layernorm(ops, dev, dtype, rows=9, cols=512, act="gelu"):
    """y = LN(act(x)) and its backward (the torchscale FFN's gelu -> ffn_layernorm pair), wide rows included."""
    fn = {"gelu": oops.gelu_erf, "quick_gelu": oops.quick_gelu}[act]
    x = rnd((rows, cols), 101, 1.5)
    g = 1 + 0.1 * rnd((cols,), 102)
    b = 0.1 * rnd((cols,), 103)
    dy = rnd((rows, cols), 104)
    if dtype == BF:
        x, dy = q(x), q(dy)
    xr = x.clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = oops.layer_norm(fn(xr), gr, br, 1e-5)
    yr.backward(dy)
    y, mean, rstd = ops.layernorm_fwd(x.to(dev, dtype), g.to(dev), b.to(dev), 1e-5, act=act)
    dgam, dbet = torch.zeros(cols, device=dev), torch.zeros(cols, device=dev)
    dxs = torch.zeros(cols, device=dev)
    dx = ops.layernorm_bwd(dy.to(dev, dtype), x.to(dev, dtype), mean, rstd, g.to(dev), dgam, dbet, act=act, dxsum=dxs)
    rt, at = (2e-2, 2e-2) if dtype == BF else (1e-4, 1e-5)
    check(f"actln.{act}.dxsum", dxs, xr.grad.sum(0), 1e-3, (2e-2 if dtype == BF else 1e-4) * max(1.0, rows ** 0.5))
    check(f"actln.{act}.y", y, yr, rt, at)
    check(f"actln.{act}.dx", dx, xr.grad, rt, at)
    check(f"actln.{act}.dgamma", dgam, gr.grad, 2e-2 if dtype == BF else 1e-4, 2e-2 if dtype == BF else 1e-4)
    check(f"actln.{act}.dbeta", dbet, br.grad, 1e-4, 1e-4)


# ------------------------------------------------------------------------------ M2 feed-forward with the sub-LayerNorm folded into its GEMMs
def case_ffn_fold(ops, dev, tokens=40, d=64, ff=192, eps=1e-5, seed=300, res_scale=1.0, act="gelu"):
    """fc1 -> gelu -> ffn_layernorm -> fc2 (+ residual) and its whole backward through antmmf_ffn_* vs fp32 autograd on the same bf16-rounded
    operands (reference feedforward_network.py:117-128).  Small shapes take the element-wise epilogues + the row / column passes; 256-aligned
    ones under ANTMMF_GEMM_FORCE_TILE=k the persistent kernel's epilogues with their per-tile partial sums."""
    fn = {"gelu": oops.gelu_erf, "quick_gelu": oops.quick_gelu}[act]
    x = q(rnd((tokens, d), seed, 1.0))
    W1 = rnd((ff, d), seed + 1, d ** -0.5)
    b1 = 0.2 * rnd((ff,), seed + 2)
    gam = 1 + 0.2 * rnd((ff,), seed + 3)
    bet = 0.1 * rnd((ff,), seed + 4)
    W2 = rnd((d, ff), seed + 5, ff ** -0.5)
    b2 = 0.1 * rnd((d,), seed + 6)
    res = q(rnd((tokens, d), seed + 7, res_scale))
    dy = q(rnd((tokens, d), seed + 8, 1.0))
    # fp32 reference on the operands the kernels see (bf16 weights for the GEMMs, fp32 LayerNorm parameters)
    xr = x.clone().requires_grad_(True)
    W1r, b1r = q(W1).requires_grad_(True), b1.clone().requires_grad_(True)
    gr, br = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    W2r, b2r = W2.clone().requires_grad_(True), b2.clone().requires_grad_(True)
    ur = xr @ W1r.t() + b1r
    ur.retain_grad()
    zr = fn(ur)
    yr = oops.layer_norm(zr, gr, br, eps) @ W2r.t() + b2r + res
    yr.backward(dy)

    to = lambda t, dt=torch.float32: t.to(dev, dt)
    w2g, c, b2f = ops.ffn_prepare_w2(to(W2), to(gam), to(bet), to(b2))
    check("ffn.w2g", w2g, W2 * gam[None, :], 1e-2, 1e-2)
    check("ffn.c", c, w2g.float().cpu().sum(1), 1e-5, 1e-5)
    check("ffn.b2f", b2f, b2 + W2 @ bet, 1e-5, 1e-5)
    z, dact, stats = ops.ffn_fc1_fwd(to(x, BF), to(q(W1), BF), to(b1), act, eps)
    check("ffn.z", z, zr, 2e-2, 1e-2)
    zq = z.float().cpu()
    mu_ref, var_ref = zq.mean(1), zq.var(1, unbiased=False)
    check("ffn.mu", stats[:, 0], mu_ref, 1e-4, 1e-4)
    check("ffn.rstd", stats[:, 1], (var_ref + eps).rsqrt(), 1e-4, 1e-4)
    ug = ur.detach().clone().requires_grad_(True)
    fn(ug).sum().backward()
    check("ffn.dact", dact, ug.grad, 2e-2, 1e-2)
    y = ops.ffn_fc2_fwd(z, w2g, c, b2f, stats, to(res, BF))
    check("ffn.y", y, yr, 2e-2, 1e-2)
    # backward
    s_col, cs_col = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
    rowv4, dys = ops.ffn_bwd_rows(to(dy, BF), y, to(res, BF), b2f, c, stats, ff, s_col, cs_col)
    check("ffn.cs", cs_col, dy.sum(0), 1e-4, 1e-4)
    t_ref = dy @ w2g.float().cpu()                               # gamma (.) dn
    zhat = (zq - mu_ref[:, None]) * (var_ref[:, None] + eps).rsqrt()
    check("ffn.m1", rowv4[:, 2], t_ref.mean(1), 1e-3, 1e-3)
    check("ffn.m2", rowv4[:, 3], (t_ref * zhat).mean(1), 5e-2, 5e-2)   # from y - b2f - res in bf16: exact up to the rounding of y
    db1 = torch.zeros(ff, device=dev)
    du = ops.ffn_fc2_dgrad(to(dy, BF), ops.transpose_bf16(w2g), z, dact, rowv4, db1)
    check("ffn.du", du, ur.grad, 3e-2, 2e-2)
    check("ffn.db1", db1, du.float().cpu().sum(0), 1e-3, 1e-3 * max(1.0, tokens ** 0.5))
    check("ffn.db1.ref", db1, b1r.grad, 3e-2, 3e-2 * max(1.0, tokens ** 0.5) / 4)
    Gm = torch.zeros(d, ff, device=dev)
    ops.gemm_wgrad_(Gm, dys, z)
    dW2, dgam, dbet = torch.zeros(d, ff, device=dev), torch.zeros(ff, device=dev), torch.zeros(ff, device=dev)
    ops.ffn_wgrad_post_(dW2, Gm, to(W2), to(gam), to(bet), s_col, cs_col, dgam, dbet)
    check("ffn.dW2", dW2, W2r.grad, 3e-2, 2e-2)
    check("ffn.dgamma", dgam, gr.grad, 3e-2, 3e-2)
    check("ffn.dbeta", dbet, br.grad, 2e-2, 2e-2)


# ------------------------------------------------------------------------------ activations / l2norm / colsum / movers
def case_activations(ops, dev):
    for act, fn in (("gelu", oops.gelu_erf), ("quick_gelu", oops.quick_gelu), ("relu", torch.relu)):
        u = rnd((6, 40), 7, 2.0)
        dg = rnd((6, 40), 8)
        ur = u.clone().requires_grad_(True)
        fn(ur).backward(dg)
        check(f"act.{act}.fwd", ops.act_fwd(u.to(dev), act), fn(u), 1e-5, 1e-6)
        check(f"act.{act}.bwd", ops.act_bwd(dg.to(dev), u.to(dev), act), ur.grad, 1e-4, 1e-5)
        ub, dgb = q(u), q(dg)
        check(f"act.{act}.fwd.bf16", ops.act_fwd(ub.to(dev, BF), act), fn(ub), 2e-2, 1e-2)
        ur = ub.clone().requires_grad_(True)
        fn(ur).backward(dgb)
        check(f"act.{act}.bwd.bf16", ops.act_bwd(dgb.to(dev, BF), ub.to(dev, BF), act), ur.grad, 2e-2, 1e-2)


def case_l2norm(ops, dev):
    x = rnd((9, 70), 11)
    dy = rnd((9, 70), 12)
    xr = x.clone().requires_grad_(True)
    yr = oops.l2_normalize(xr)
    yr.backward(dy)
    y, inv = ops.l2norm_fwd(x.to(dev), 1e-12)
    check("l2.y", y, yr, 1e-5, 1e-6)
    check("l2.dx", ops.l2norm_bwd(dy.to(dev), y, inv, torch.float32), xr.grad, 1e-4, 1e-5)
    xb = q(x)
    yb, invb = ops.l2norm_fwd(xb.to(dev, BF), 1e-12, out_dtype=torch.float32)  # bf16 tower output -> fp32 embedding
    check("l2.y.bf16in", yb, oops.l2_normalize(xb), 1e-5, 1e-6)
    dxb = ops.l2norm_bwd(dy.to(dev), yb, invb, BF)
    xr = xb.clone().requires_grad_(True)
    oops.l2_normalize(xr).backward(dy)
    check("l2.dx.bf16", dxb, xr.grad, 2e-2, 1e-2)


def case_movers(ops, dev):
    x = q(rnd((37, 24), 13))
    out = torch.full((24,), 0.5, device=dev)
    ops.colsum_(out, x.to(dev, BF))
    check("colsum", out, x.sum(0) + 0.5, 1e-5, 1e-5)
    big = q(rnd((40, 48), 14))
    view = big.to(dev, BF)[:, 8:24]  # strided view
    out = torch.zeros(16, device=dev)
    ops.colsum_(out, view)
    check("colsum.strided", out, big[:, 8:24].sum(0), 1e-5, 1e-5)
    w = q(rnd((70, 130), 15))
    check("transpose", ops.transpose_bf16(w.to(dev, BF)), w.t(), 0, 0)
    f = rnd((1000,), 16)
    check("cast", ops.cast_bf16(f.to(dev)), q(f), 0, 0)
    # patchify + assemble + split
    img = rnd((2, 3, 16, 16), 17)
    from oracle.towers import patchify as opatch

    pt = ops.patchify(img.to(dev), 4, kpad=64, shift=0.5, scale=2.0)
    ref = q((opatch(img, 4) - 0.5) * 2.0).reshape(-1, 48)
    check("patchify", pt[:, :48], ref, 0, 0)
    assert float(pt[:, 48:].abs().max()) == 0.0
    d = 32
    patch_tokens = q(rnd((2 * 16, d), 18))
    cls, pos, bias = rnd((d,), 19), rnd((17, d), 20), rnd((d,), 21)
    tok = ops.assemble_tokens(patch_tokens.to(dev, BF), cls.to(dev), pos.to(dev), bias.to(dev), 2, 16)
    ref = torch.cat([cls.expand(2, 1, d), patch_tokens.view(2, 16, d) + bias], 1) + pos
    check("assemble", tok, q(ref), 0, 1e-6)
    check("split", ops.split_tokens(tok), tok[:, 1:].reshape(-1, d), 0, 0)
    # embedding gather / scatter
    ids = torch.randint(0, 50, (3, 5), generator=torch.Generator().manual_seed(1))
    word, posT, typ = rnd((50, d), 22), rnd((12, d), 23), rnd((2, d), 24)
    zero = torch.zeros(3, 5, dtype=torch.uint8)
    zero[1, 3:] = 1
    e = ops.embed_gather(ids.to(dev), word.to(dev), posT.to(dev), typ.to(dev), None, zero.to(dev), pos_offset=2)
    ref = (word[ids] + posT[2:7][None] + typ[0]) * (1 - zero[..., None].float())
    check("embed_gather", e, q(ref), 0, 1e-6)
    dx = q(rnd((3, 5, d), 25))
    dword = torch.zeros(50, d, device=dev)
    ops.embed_scatter_add_(dword, dx.to(dev, BF), ids.to(dev), zero.to(dev))
    ref = torch.zeros(50, d).index_add_(0, ids.flatten(), (dx * (1 - zero[..., None].float())).reshape(-1, d))
    check("embed_scatter", dword, ref, 1e-5, 1e-5)
    dpos = torch.zeros(12, d, device=dev)
    ops.embed_scatter_add_(dpos, dx.to(dev, BF), None, None, seq=5, offset=2)
    ref = torch.zeros(12, d)
    ref[2:7] = dx.sum(0)
    check("embed_scatter.pos", dpos, ref, 1e-5, 1e-5)
    # the batched position path (>= 64 sequences: a thread walks a slice of the batch per (position, vector), one atomic per slice), with skipped rows
    nb, sq = 70, 6
    dxb = q(rnd((nb, sq, 16), 38))
    skipb = (torch.arange(nb * sq) % 5 == 3).to(torch.uint8).view(nb, sq)
    dposb = torch.full((10, 16), 0.5, device=dev)
    ops.embed_scatter_add_(dposb, dxb.to(dev, BF), None, skipb.to(dev), seq=sq, offset=3)
    refb = torch.full((10, 16), 0.5)
    refb[3:3 + sq] += (dxb * (1 - skipb.float())[..., None]).sum(0)
    check("embed_scatter.pos.batched", dposb, refb, 1e-5, 1e-5)
    # the sorted word-table path (>= 8192 rows: one writer per table row, no atomics), repeated ids, skipped rows, untouched rows
    nbw, sqw, V = 1100, 8, 37
    idsw = torch.randint(0, V - 3, (nbw, sqw), generator=torch.Generator().manual_seed(3))
    dxw = q(rnd((nbw, sqw, 16), 39))
    skipw = (torch.arange(nbw * sqw) % 7 == 2).to(torch.uint8).view(nbw, sqw)
    dw2 = torch.full((V, 16), -0.25, device=dev)
    ops.embed_scatter_add_(dw2, dxw.to(dev, BF), idsw.to(dev), skipw.to(dev))
    refw = torch.full((V, 16), -0.25).index_add_(0, idsw.flatten(), (dxw * (1 - skipw.float())[..., None]).reshape(-1, 16))
    check("embed_scatter.sorted", dw2, refw, 1e-5, 1e-5)


def case_adamw(ops, dev):
    n = 1000
    p, g = rnd((n,), 30), rnd((n,), 31)
    ref_p = torch.nn.Parameter(p.clone())
    opt = torch.optim.AdamW([ref_p], lr=1e-2, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.05)
    pd, m, v = p.clone().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    shadow = torch.empty(n, dtype=BF, device=dev)
    for step in (1, 2, 3):
        ref_p.grad = g.clone() * step
        opt.step()
        ops.adamw_step_(pd, (g * step * 4).to(dev), m, v, shadow, 1e-2, 0.9, 0.98, 1e-6, 0.05, step, grad_scale=0.25)
    check("adamw.p", pd, ref_p.data, 1e-5, 1e-6)
    check("adamw.shadow", shadow, q(pd.cpu()), 0, 0)
    s = torch.zeros(1, device=dev)
    ops.sumsq_(s, g.to(dev))
    check("sumsq", s, (g * g).sum()[None], 1e-5, 1e-6)


# ------------------------------------------------------------------------------ GEMM
def case_gemm(ops, dev, I=70, J=40, R=72):
    X = q(rnd((I, R), 40))
    Wt = q(rnd((J, R), 41, R ** -0.5))
    bias = rnd((J,), 42)
    res = q(rnd((I, J), 43))
    Xd, Wd = X.to(dev, BF), Wt.to(dev, BF)
    pre = X @ Wt.t() + bias
    # forward, bias + quick_gelu + residual + aux
    aux = torch.empty(I, J, dtype=BF, device=dev)
    y = ops.gemm(Xd, Wd, bias=bias.to(dev), act="quick_gelu", residual=res.to(dev, BF), aux=aux)
    check("gemm.nt.aux", aux, pre, 2e-2, 1e-2)
    check("gemm.nt.fused", y, oops.quick_gelu(pre) + res, 2e-2, 1e-2)
    # fp32 output with alpha, accumulate
    c = torch.ones(I, J, device=dev)
    ops.gemm(Xd, Wd, out=c, alpha=0.5, accumulate=True)
    check("gemm.nt.f32.acc", c, 1 + 0.5 * (X @ Wt.t()), 1e-3, 1e-3)
    # dgrad: dX = dY W  (P = dY [I, J], Q = W [J, R] r-major, reduction over J), gated by act'(u)
    dY = q(rnd((I, J), 44))
    u = q(rnd((I, R), 45))
    dx = ops.gemm(dY.to(dev, BF), Wd, q_rmajor=True, gate=u.to(dev, BF), act="gelu")
    ur = u.clone().requires_grad_(True)
    oops.gelu_erf(ur).backward(dY @ Wt)
    check("gemm.nn.gate", dx, ur.grad, 2e-2, 1e-2)
    # the same pair with the derivative handed over instead of the pre-activation: forward aux = act'(pre), dgrad gate = that derivative
    for act, fn in (("quick_gelu", oops.quick_gelu), ("gelu", oops.gelu_erf)):
        aux2 = torch.empty(I, J, dtype=BF, device=dev)
        y2 = ops.gemm(Xd, Wd, bias=bias.to(dev), act=act, aux=aux2, aux_grad=True)
        pr = pre.clone().requires_grad_(True)
        fn(pr).backward(torch.ones_like(pre))
        check(f"gemm.nt.aux_grad.{act}", aux2, pr.grad, 2e-2, 1e-2)
        check(f"gemm.nt.aux_grad.{act}.y", y2, fn(pre), 2e-2, 1e-2)
        d = q(rnd((I, R), 46)) * 0.5 + 0.5
        dx2 = ops.gemm(dY.to(dev, BF), Wd, q_rmajor=True, gate=d.to(dev, BF), act=act, gate_is_grad=True)
        check(f"gemm.nn.gate_is_grad.{act}", dx2, (dY @ Wt) * d, 2e-2, 1e-2)
    # wgrad: dW = dY^T X  (both r-major, reduction over tokens), fp32 accumulate, with and without split-k
    for split in (1, 2):
        dw = torch.full((J, R), 0.25, device=dev)
        ops.gemm(dY.to(dev, BF), Xd, out=dw, p_rmajor=True, q_rmajor=True, accumulate=True, split_k=split)
        check(f"gemm.tn.split{split}", dw, 0.25 + dY.t() @ X, 1e-3, 1e-3)
    # strided views of a packed buffer (ld != width), output into a column slice
    packed = torch.zeros(I, 3 * J, dtype=BF, device=dev)
    ops.gemm(Xd, Wd, out=packed[:, J:2 * J])
    check("gemm.nt.ldc", packed[:, J:2 * J], X @ Wt.t(), 2e-2, 1e-2)
    assert float(packed[:, :J].float().abs().max()) == 0.0 and float(packed[:, 2 * J:].float().abs().max()) == 0.0


def case_gemm_multitile(ops, dev):
    """> 1 tile in every direction, K tail (R % 64 != 0), asymmetric operands (catches transposed C)."""
    I, J, R = 150, 136, 200
    X = q(rnd((I, R), 46))
    Wt = q(rnd((J, R), 47, R ** -0.5))
    y = ops.gemm(X.to(dev, BF), Wt.to(dev, BF), out_dtype=torch.float32)
    check("gemm.multitile", y, X @ Wt.t(), 1e-3, 1e-3)
    dY = q(rnd((I, J), 48))
    dw = torch.zeros(J, R, device=dev)
    ops.gemm(dY.to(dev, BF), X.to(dev, BF), out=dw, p_rmajor=True, q_rmajor=True, accumulate=True)
    check("gemm.multitile.tn", dw, dY.t() @ X, 1e-3, 1e-3)
    dx = ops.gemm(dY.to(dev, BF), Wt.to(dev, BF), q_rmajor=True, out_dtype=torch.float32)
    check("gemm.multitile.nn", dx, dY @ Wt, 1e-3, 1e-3)
    # R % 64 == 0 with both operands r-contiguous takes the LDS-DMA kernel (row clamping at the ragged edges)
    I, J, R = 150, 136, 192
    X = q(rnd((I, R), 49))
    Wt = q(rnd((J, R), 50, R ** -0.5))
    bias = rnd((J,), 51)
    res = q(rnd((I, J), 52))
    y = ops.gemm(X.to(dev, BF), Wt.to(dev, BF), bias=bias.to(dev), act="gelu", residual=res.to(dev, BF))
    check("gemm.dma.fused", y, oops.gelu_erf(X @ Wt.t() + bias) + res, 2e-2, 1e-2)
    y = ops.gemm(X.to(dev, BF), Wt.to(dev, BF), out_dtype=torch.float32)
    check("gemm.dma.f32", y, X @ Wt.t(), 1e-3, 1e-3)
    # wgrad with tokens % 64 == 0 and 128-aligned outputs takes the LDS-DMA + transpose-read kernel
    tokens, n_out, k_in = 192, 128, 256
    dY = q(rnd((tokens, n_out), 53))
    Xa = q(rnd((tokens, k_in), 54))
    for split in (1, 3):
        dw = torch.full((n_out, k_in), -0.5, device=dev)
        ops.gemm(dY.to(dev, BF), Xa.to(dev, BF), out=dw, p_rmajor=True, q_rmajor=True, accumulate=True, split_k=split)
        check(f"gemm.tn.dma.split{split}", dw, dY.t() @ Xa - 0.5, 1e-3, 1e-3)


def case_gemm_persistent(ops, dev, I=700, J=600, R=192, quick=False):
    """Forward-layout GEMM big enough that workgroups of the persistent ring kernel walk several 256 x 256 tiles (ragged edges,
    every compile-time epilogue; quick: two of them).  R must be a multiple of 64 and >= 128 for the dispatcher to pick that kernel."""
    X = q(rnd((I, R), 146))
    Wt = q(rnd((J, R), 147, R ** -0.5))
    bias = rnd((J,), 148)
    res = q(rnd((I, J), 149))
    ref = X @ Wt.t()
    Xd, Wd = X.to(dev, BF), Wt.to(dev, BF)
    check("gemm.persist.plain", ops.gemm(Xd, Wd), ref, 2e-2, 1e-2)
    check("gemm.persist.bias_res", ops.gemm(Xd, Wd, bias=bias.to(dev), residual=res.to(dev, BF)), ref + bias + res, 2e-2, 1e-2)
    if quick:
        return
    check("gemm.persist.bias", ops.gemm(Xd, Wd, bias=bias.to(dev)), ref + bias, 2e-2, 1e-2)
    check("gemm.persist.res", ops.gemm(Xd, Wd, residual=res.to(dev, BF)), ref + res, 2e-2, 1e-2)
    check("gemm.persist.generic", ops.gemm(Xd, Wd, bias=bias.to(dev), act="gelu"), oops.gelu_erf(ref + bias), 2e-2, 1e-2)


def case_gemm_k64(ops, dev, I=512, J=512, R=192, quick=False):
    """Forward-layout GEMM on the BK = 64 quarter-phase kernel (256-aligned I / J, R % 64 == 0): every compile-time epilogue and the
    generic one (aux store, activation, gate, alpha, residual) through the register-level (permlane swap) store path."""
    X = q(rnd((I, R), 246))
    Wt = q(rnd((J, R), 247, R ** -0.5))
    bias = rnd((J,), 248)
    res = q(rnd((I, J), 249))
    ref = X @ Wt.t()
    Xd, Wd = X.to(dev, BF), Wt.to(dev, BF)
    check("gemm.k64.plain", ops.gemm(Xd, Wd), ref, 2e-2, 1e-2)
    check("gemm.k64.bias_res", ops.gemm(Xd, Wd, bias=bias.to(dev), residual=res.to(dev, BF)), ref + bias + res, 2e-2, 1e-2)
    if quick:
        return
    check("gemm.k64.bias", ops.gemm(Xd, Wd, bias=bias.to(dev)), ref + bias, 2e-2, 1e-2)
    check("gemm.k64.res", ops.gemm(Xd, Wd, residual=res.to(dev, BF)), ref + res, 2e-2, 1e-2)
    aux = torch.empty(I, J, dtype=BF, device=dev)
    y = ops.gemm(Xd, Wd, bias=bias.to(dev), act="quick_gelu", residual=res.to(dev, BF), aux=aux, alpha=0.5)
    pre = 0.5 * ref + bias
    check("gemm.k64.generic.aux", aux, pre, 2e-2, 1e-2)
    check("gemm.k64.generic", y, oops.quick_gelu(pre) + res, 2e-2, 1e-2)
    u = q(rnd((I, J), 250))
    dx = ops.gemm(Xd, Wd, gate=u.to(dev, BF), act="gelu")
    ur = u.clone().requires_grad_(True)
    oops.gelu_erf(ur).backward(ref)
    check("gemm.k64.gate", dx, ur.grad, 2e-2, 1e-2)
    aux2 = torch.empty(I, J, dtype=BF, device=dev)
    y2 = ops.gemm(Xd, Wd, bias=bias.to(dev), act="gelu", aux=aux2, aux_grad=True)
    pr = (ref + bias).clone().requires_grad_(True)
    oops.gelu_erf(pr).backward(torch.ones_like(ref))
    check("gemm.k64.aux_grad", aux2, pr.grad, 2e-2, 1e-2)
    check("gemm.k64.aux_grad.y", y2, oops.gelu_erf(ref + bias), 2e-2, 1e-2)
    dx2 = ops.gemm(Xd, Wd, gate=u.to(dev, BF), act="gelu", gate_is_grad=True)
    check("gemm.k64.gate_is_grad", dx2, ref * u, 2e-2, 1e-2)
    if ops.gemm_gated_colsum_ok(I, J, R):
        # the same gated dgrad with its column sums as 256 partial rows (the bias gradient of the Linear in front of the activation): the product bit-identical,
        # the partial rows' column sums = the column sums of the fp32 product
        dx3, parts = ops.gemm_gated_colsum(Xd, Wd, u.to(dev, BF))
        assert torch.equal(dx3, dx2), "gemm.k64.gated_colsum: product differs"
        assert tuple(parts.shape) == (256, J)
        check("gemm.k64.gated_colsum", parts.sum(0), (ref * u).sum(0), 2e-2, 1e-2)
    packed = torch.zeros(I, 2 * J, dtype=BF, device=dev)
    ops.gemm(Xd, Wd, out=packed[:, J:])
    check("gemm.k64.ldc", packed[:, J:], ref, 2e-2, 1e-2)
    assert float(packed[:, :J].float().abs().max()) == 0.0


def case_gemm_wgrad_ring(ops, dev, tokens=4096, n_out=256, k_in=256, quick=False):
    """wgrad through the 4-stage DMA ring + transpose reads (256-aligned outputs, >= 4096 tokens), split over tokens."""
    dY = q(rnd((tokens, n_out), 55, 0.5))
    Xa = q(rnd((tokens, k_in), 56, 0.5))
    dw = torch.full((n_out, k_in), 0.75, device=dev)
    ops.gemm(dY.to(dev, BF), Xa.to(dev, BF), out=dw, p_rmajor=True, q_rmajor=True, accumulate=True)   # no workspace: fp32 atomics
    check("gemm.tn.ring.atomic", dw, dY.t() @ Xa + 0.75, 2e-3, 2e-3)
    dw = torch.full((n_out, k_in), 0.75, device=dev)
    ops.gemm_wgrad_(dw, dY.to(dev, BF), Xa.to(dev, BF))                                             # workspace + reduce launch
    check("gemm.tn.ring.ws", dw, dY.t() @ Xa + 0.75, 2e-3, 2e-3)
    if quick:   # (the CPU lane emulator runs the first two forms; the strided destination runs on hardware)
        return
    big = torch.zeros(n_out, k_in + 64, device=dev)
    ops.gemm_wgrad_(big[:, :k_in], dY.to(dev, BF), Xa.to(dev, BF))                                  # strided destination
    check("gemm.tn.ring.ws.ld", big[:, :k_in], dY.t() @ Xa, 2e-3, 2e-3)
    assert float(big[:, k_in:].abs().max()) == 0.0


def case_gemm_wgrad_seg(ops, dev, tokens=4096, rows=256, k_in=256, n_seg=3):
    """One wgrad GEMM over the packed dQ | dK | dV whose reduce launch scatters the row segments into separate gradient buffers (antmmf_gemm_wgrad_bf16_seg) against
    per-segment calls and the fp32 product; a shape off the workspace path (tokens < 4096) takes the per-segment fallback inside the entry point."""
    dY = q(rnd((tokens, n_seg * rows), 57, 0.5))
    Xa = q(rnd((tokens, k_in), 58, 0.5))
    # the three destinations at unrelated offsets of one buffer, in an order that is NOT the column order (the arena holds k, v, q)
    arena = torch.full((n_seg * rows * k_in + 3 * 64,), 0.25, device=dev)
    order = [(s_ + 1) % n_seg for s_ in range(n_seg)]
    dst = [arena[order[s_] * (rows * k_in + 64):order[s_] * (rows * k_in + 64) + rows * k_in].view(rows, k_in) for s_ in range(n_seg)]
    ops.gemm_wgrad_seg_(dst, dY.to(dev, BF), Xa.to(dev, BF))
    for s_ in range(n_seg):
        check(f"gemm.tn.seg.{s_}", dst[s_], dY[:, s_ * rows:(s_ + 1) * rows].t() @ Xa + 0.25, 2e-3, 2e-3)
    pads = torch.cat([arena[i * (rows * k_in + 64) + rows * k_in:(i + 1) * (rows * k_in + 64)] for i in range(n_seg)])
    assert float((pads - 0.25).abs().max()) == 0.0          # nothing outside the segments was touched
    small = [torch.zeros(rows, k_in, device=dev) for _ in range(n_seg)]
    ops.gemm_wgrad_seg_(small, dY[:1024].to(dev, BF), Xa[:1024].to(dev, BF))
    for s_ in range(n_seg):
        check(f"gemm.tn.seg.fallback.{s_}", small[s_], dY[:1024, s_ * rows:(s_ + 1) * rows].t() @ Xa[:1024], 2e-3, 2e-3)


# ------------------------------------------------------------------------------ attention
def _attn_ref(qh, kh, vh, scale, key_bias):
    return oops.attention_core(qh, kh, vh, scale, key_bias)


def case_attention(ops, dev, B=2, heads=2, Nq=17, Nk=17, bias_kind="none", packed=True, seed=50, head_dim=64):
    D = heads * head_dim
    if packed and Nq == Nk:
        qkv = q(rnd((B, Nq, 3 * D), seed))
        qt, kt, vt = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        dqkv = qkv.to(dev, BF)
        qd, kd, vd = dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:]
    else:
        qt, kt, vt = q(rnd((B, Nq, D), seed)), q(rnd((B, Nk, D), seed + 1)), q(rnd((B, Nk, D), seed + 2))
        qd, kd, vd = qt.to(dev, BF), kt.to(dev, BF), vt.to(dev, BF)
    key_bias = None
    if bias_kind != "none":
        lengths = torch.tensor([Nk, max(1, Nk // 3)] * B)[:B]
        valid = torch.arange(Nk)[None, :] < lengths[:, None]
        key_bias = torch.zeros(B, Nk).masked_fill(~valid, -10000.0 if bias_kind == "bert" else float("-inf"))
    scale = head_dim ** -0.5
    d_o = q(rnd((B, Nq, D), seed + 3))
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (qt, kt, vt))
    ref = oops.merge_heads(_attn_ref(oops.split_heads(qr, heads), oops.split_heads(kr, heads), oops.split_heads(vr, heads), scale, key_bias))
    ref.backward(d_o)
    kb = None if key_bias is None else key_bias.to(dev)
    o, lse = ops.attention_fwd(qd, kd, vd, heads, scale, kb)
    tag = f"attn[{B},{heads}x{head_dim},{Nq},{Nk},{bias_kind}]"
    check(tag + ".o", o, ref, 2e-2, 2e-2)
    s = torch.matmul(oops.split_heads(qt, heads), oops.split_heads(kt, heads).transpose(-1, -2)) * scale
    if key_bias is not None:
        s = s + key_bias[:, None, None, :]
    check(tag + ".lse", lse, torch.logsumexp(s, -1), 1e-2, 1e-2)
    dq, dk, dv = ops.attention_bwd(qd, kd, vd, o, lse, d_o.to(dev, BF), heads, scale, kb)
    check(tag + ".dq", dq, qr.grad, 3e-2, 3e-2)
    check(tag + ".dk", dk, kr.grad, 3e-2, 3e-2)
    check(tag + ".dv", dv, vr.grad, 3e-2, 3e-2)
    if head_dim == 64 and ops.attention_bwd_sums_ok(head_dim, Nq, Nk):
        # the same backward with the per-batch-item token sums of dQ | dK | dV (the q / k / v bias gradients' raw material): gradients bit-identical to the
        # plain call, sums = fp32 sums of the unrounded gradients -- against the oracle's gradients and against the sums of the bf16 outputs
        sums = torch.full((B, 3 * D), float("nan"), dtype=torch.float32, device=dev)
        dq2, dk2, dv2 = ops.attention_bwd(qd, kd, vd, o, lse, d_o.to(dev, BF), heads, scale, kb, sums=sums)
        assert torch.equal(dq2, dq) and torch.equal(dk2, dk) and torch.equal(dv2, dv), tag + ": gradients differ with sums requested"
        ref_s = torch.cat([qr.grad.sum(1), kr.grad.sum(1), vr.grad.sum(1)], dim=1)
        own_s = torch.cat([dq.float().sum(1), dk.float().sum(1), dv.float().sum(1)], dim=1)
        check(tag + ".sums", sums, ref_s, 3e-2, 3e-2)
        check(tag + ".sums_vs_outputs", sums, own_s, 2e-2, 2e-2)
        sums2 = torch.full((B, 3 * D), 7.0, dtype=torch.float32, device=dev)   # without the dV third: left as it was, the rest the same bits
        dq3, dk3, dv3 = ops.attention_bwd(qd, kd, vd, o, lse, d_o.to(dev, BF), heads, scale, kb, sums=sums2, sums_v=False)
        assert torch.equal(dq3, dq) and torch.equal(dk3, dk) and torch.equal(dv3, dv)
        assert torch.equal(sums2[:, :2 * D], sums[:, :2 * D]) and bool((sums2[:, 2 * D:] == 7.0).all()), tag + ": sums without dV"


# ------------------------------------------------------------------------------ losses
def case_milnce(ops, dev, Bg=6, n=2, world=2):
    """Row-sharded MIL-NCE over `world` simulated ranks == oracle loss on the tiled global matrix, and the
    slab gradients reproduce d loss / d (text, clip) embeddings."""
    D = 16
    T = rnd((Bg, D), 60, 0.7).requires_grad_(True)
    V = rnd((Bg * n, D), 61, 0.7).requires_grad_(True)
    simi = torch.matmul(V.view(Bg, n, D), T.t()).permute(2, 0, 1)
    mil = simi.unsqueeze(1).expand(Bg, n, Bg, n).reshape(Bg * n, Bg * n)
    wv = rnd((Bg,), 62).abs() + 0.2
    ref = olosses.mil_nce(mil, Bg, n, wv)
    ref.backward()
    Bl = Bg // world
    total = 0.0
    dT, dV = torch.zeros(Bg, D), torch.zeros(Bg * n, D)
    Td, Vd = T.detach(), V.detach()
    for r in range(world):
        lo = r * Bl
        Rm = (Td[lo:lo + Bl] @ Vd.t()).contiguous()
        Vc = Vd.view(Bg, n, D)[lo:lo + Bl, n // 2]
        Cm = (Vc @ Td.t()).contiguous()
        loss_rows, denom = ops.milnce_fwd(Rm.to(dev), Cm.to(dev), n, lo)
        coef = wv[lo:lo + Bl] / Bg
        total = total + float((loss_rows.cpu() * coef).sum())
        dR, dC = ops.milnce_bwd(Rm.to(dev), Cm.to(dev), denom, coef.to(dev), n, lo, out_dtype=torch.float32)
        dR, dC = dR.cpu(), dC.cpu()
        dT[lo:lo + Bl] += dR @ Vd
        dV += dR.t() @ Td[lo:lo + Bl]
        dV.view(Bg, n, D)[lo:lo + Bl, n // 2] += dC @ Td
        dT += dC.t() @ Vc
    assert abs(total - float(ref)) <= 1e-5 * max(1.0, abs(float(ref))), (total, float(ref))
    check("milnce.dT", dT, T.grad, 1e-4, 1e-5)
    check("milnce.dV", dV, V.grad, 1e-4, 1e-5)


def case_softmax_ce(ops, dev, Bg=8, world=2):
    D = 16
    Im = oops.l2_normalize(rnd((Bg, D), 70)).requires_grad_(True)
    Tx = oops.l2_normalize(rnd((Bg, D), 71)).requires_grad_(True)
    ls = torch.tensor(math.log(1 / 0.07), requires_grad=True)
    ref, _ = olosses.clip_itc(Im, Tx, ls)
    ref.backward()
    Bl = Bg // world
    total, dI, dTx, dls = 0.0, torch.zeros(Bg, D), torch.zeros(Bg, D), torch.zeros(1, device=dev)
    Id, Td = Im.detach(), Tx.detach()
    lsd = ls.detach().reshape(1).to(dev)
    for r in range(world):
        lo = r * Bl
        for A, Bm, dA, dB in ((Id, Td, dI, dTx), (Td, Id, dTx, dI)):
            x = (A[lo:lo + Bl] @ Bm.t()).contiguous()
            loss_rows, lse = ops.softmax_ce_fwd(x.to(dev), lo, lsd)
            coef = torch.full((Bl,), 0.5 / Bg)
            total += float((loss_rows.cpu() * coef).sum())
            dx = ops.softmax_ce_bwd(x.to(dev), lse, coef.to(dev), lo, lsd, dscale=dls, out_dtype=torch.float32).cpu()
            dA[lo:lo + Bl] += dx @ Bm
            dB += dx.t() @ A[lo:lo + Bl]
    assert abs(total - float(ref)) <= 1e-5 * max(1.0, abs(float(ref))), (total, float(ref))
    check("ce.dI", dI, Im.grad, 1e-4, 1e-5)
    check("ce.dT", dTx, Tx.grad, 1e-4, 1e-5)
    check("ce.dlogscale", dls.cpu() * float(ls.exp()), ls.grad.reshape(1), 1e-4, 1e-5)
    # CrossEn (scale 100, no learnable scale)
    s = rnd((7, 7), 72, 0.02).requires_grad_(True)
    refc = olosses.cross_en(s)
    refc.backward()
    lr, lse = ops.softmax_ce_fwd(s.detach().to(dev), 0, None, 100.0)
    assert abs(float(lr.mean()) - float(refc)) < 1e-5 * max(1.0, abs(float(refc)))
    dx = ops.softmax_ce_bwd(s.detach().to(dev), lse, torch.full((7,), 1 / 7.).to(dev), 0, None, 100.0, out_dtype=torch.float32)
    check("crossen.dx", dx, s.grad, 1e-4, 1e-5)


def case_moco(ops, dev, R=7, Np=2, K=300):
    """Fused MoCo loss rows + gradients vs the oracle (MocoUtils.moco_loss), and the momentum-update kernel."""
    pos = rnd((R, Np), 301, 0.3).requires_grad_(True)
    neg = rnd((R, K), 302, 0.3).requires_grad_(True)
    ref = olosses.moco(pos, neg, 0.05)
    ref.backward()
    rows, la, lp = ops.moco_fwd(pos.detach().to(dev), neg.detach().to(dev), 0.05)
    check("moco.loss", rows.mean(), ref.detach(), 1e-5, 1e-6)
    coef = torch.full((R,), 1.0 / R, device=dev)
    dpos, dneg = ops.moco_bwd(pos.detach().to(dev), neg.detach().to(dev), la, lp, coef, 0.05, out_dtype=torch.float32)
    check("moco.dpos", dpos, pos.grad, 1e-4, 1e-5)
    check("moco.dneg", dneg, neg.grad, 1e-4, 1e-5)
    n = 1000 + 3
    k, qv = rnd((n,), 303), rnd((n,), 304)
    want = k * 0.9 + qv * 0.1
    kd = k.clone().to(dev)
    sh = torch.zeros(n, dtype=BF, device=dev)
    ops.ema_update_(kd, qv.to(dev), 0.9, sh)
    check("ema.k", kd, want, 1e-6, 1e-6)
    check("ema.shadow", sh, want, 1e-2, 1e-2)


def case_dmae_losses(contrastive, dev, B=9):
    """CrossEn / NegNCE (DMAE) autograd wrappers over the fused row kernels vs the oracle restatements (pinned to the reference
    in tests/test_oracle_golden.py::test_loss_misc)."""
    for seed, sc in ((401, 0.05), (402, 0.01)):
        S = (rnd((B, B), seed, sc) + 0.02 * torch.eye(B)).requires_grad_(True)
        for name, ofn, fn in (("crossen", olosses.cross_en, contrastive.cross_en), ("negnce", olosses.neg_nce, contrastive.neg_nce)):
            S.grad = None
            ref = ofn(S)
            ref.backward()
            Sd = S.detach().clone().to(dev).requires_grad_(True)
            got = fn(Sd)
            got.backward()
            check(f"dmae.{name}.loss", got, ref.detach(), 1e-4, 1e-5)
            check(f"dmae.{name}.grad", Sd.grad, S.grad, 2e-3, 1e-4)


def case_tpmcl_ops(dev, C=37, V=13, T=30, D=96):
    """The DMAE stage-3 head kernels (csrc/tpmcl.hip) through their autograd wrappers vs plain torch fp32 on the same inputs: token weights
    (Linear(D, 1) + masked softmax; reference dmae_utils.py:147-165), aligned-pair dots / weighted sums (the two einsums of
    wti_interaction_row and the global-feature prediction), the TokenImportanceSelector mask vs its sort / cumsum / scatter form
    (tpmcl_utils.py:101-121), and linear_f32 (hi / lo split GEMM) forward and both gradients."""
    from antmmf.hip import tpmcl

    feat = rnd((C, T, D), 701, 1.0)
    w, b = rnd((1, D), 702, 0.3), rnd((1,), 703, 0.1)
    mask = (rnd((C, T), 704, 1.0) > -0.5).float()
    mask[:, 0] = 1.0
    dout = rnd((C, T), 705, 1.0)
    fr, wr, br = (x.clone().requires_grad_(True) for x in (feat, w, b))
    ref = torch.softmax((fr @ wr.t()).squeeze(-1).add(br).masked_fill(mask < 0.5, float("-inf")), dim=-1)
    (ref * dout).sum().backward()
    fd, wd, bd = (x.clone().to(dev).requires_grad_(True) for x in (feat, w, b))
    got = tpmcl.token_weights(fd, wd, bd, mask.to(dev))
    (got * dout.to(dev)).sum().backward()
    check("tpm.token_weights", got, ref.detach(), 1e-5, 1e-6)
    check("tpm.token_weights.dfeat", fd.grad, fr.grad, 1e-4, 1e-6)
    check("tpm.token_weights.dw", wd.grad, wr.grad, 1e-4, 1e-5)
    # (the bias gradient vanishes analytically -- a softmax does not see a shift of its logits: both sides hold rounding noise)
    assert float(bd.grad.abs().max()) <= 1e-5 * float(wr.grad.abs().max()), (float(bd.grad), float(br.grad))
    assert float(got[mask.to(dev) < 0.5].abs().max()) == 0.0

    x, y, ww = rnd((C, D), 706, 1.0), rnd((C, V, D), 707, 1.0), rnd((C, V), 708, 1.0)
    g1, g2 = rnd((C, V), 709, 1.0), rnd((C, D), 710, 1.0)
    xr, yr, wwr = (t.clone().requires_grad_(True) for t in (x, y, ww))
    dref = torch.einsum("cd,cvd->cv", xr, yr)
    sref = torch.einsum("cv,cvd->cd", wwr, yr)
    ((dref * g1).sum() + (sref * g2).sum()).backward()
    xd, yd, wwd = (t.clone().to(dev).requires_grad_(True) for t in (x, y, ww))
    dgot, sgot = tpmcl.pair_dots(xd, yd), tpmcl.pair_wsum(wwd, yd)
    ((dgot * g1.to(dev)).sum() + (sgot * g2.to(dev)).sum()).backward()
    check("tpm.pair_dots", dgot, dref.detach(), 1e-5, 1e-5)
    check("tpm.pair_wsum", sgot, sref.detach(), 1e-5, 1e-5)
    check("tpm.pair.dx", xd.grad, xr.grad, 1e-5, 1e-5)
    check("tpm.pair.dy", yd.grad, yr.grad, 1e-5, 1e-5)
    check("tpm.pair.dw", wwd.grad, wwr.grad, 1e-5, 1e-5)

    for thresh in (0.6, 0.3, 0.95):
        wts = torch.softmax(rnd((C, V), 711, 2.0), dim=-1)
        ws, order = wts.sort(dim=1, descending=True)
        keep_ref = 1.0 - torch.zeros_like(wts).scatter(1, order, (ws.cumsum(dim=1) < thresh).float())
        keep = tpmcl.tis_keep(wts.to(dev), thresh)
        assert torch.equal(keep.cpu(), keep_ref), (thresh, (keep.cpu() != keep_ref).sum())

    X, Wm = rnd((3, 11, 40), 712, 1.0), rnd((24, 40), 713, 0.2)
    bias = rnd((24,), 714, 0.1)
    G = rnd((3, 11, 24), 715, 1.0)
    Xr, Wr = X.clone().requires_grad_(True), Wm.clone().requires_grad_(True)
    (torch.nn.functional.linear(Xr, Wr, bias) * G).sum().backward()
    Xd, Wd = X.clone().to(dev).requires_grad_(True), Wm.clone().to(dev).requires_grad_(True)
    out = tpmcl.linear_f32(Xd, Wd, bias.to(dev))
    (out * G.to(dev)).sum().backward()
    check("tpm.linear_f32", out, torch.nn.functional.linear(X, Wm, bias), 2e-5, 2e-5)   # hi / lo split: fp32-class accuracy out of bf16 MFMAs
    check("tpm.linear_f32.dx", Xd.grad, Xr.grad, 2e-5, 2e-5)
    check("tpm.linear_f32.dw", Wd.grad, Wr.grad, 2e-5, 2e-5)


def case_retrieval_metrics(dev, golden):
    """GlobalRetrievalRecall's rank kernel + reductions vs the reference metric run (metric_recall.pt): square matrix with the
    diagonal as ground truth, and a 17 x 11 matrix with explicit (multi-)ground-truth lists in both directions."""
    from antmmf.modules.metrics import global_retrieval_recall as grr

    g = golden("metric_recall.pt")
    got = grr._cal_recall(g["sq.sim"].to(dev))
    for k in ("mr", "r@1", "r@5", "r@10"):
        assert abs(got[k] - float(g["sq." + k])) < 1e-6, (k, got[k], float(g["sq." + k]))
    T, V = g["rect.sim"].shape
    t2v = [[i % V] for i in range(T)]
    v2t = [[i for i in range(T) if i % V == j] for j in range(V)]
    got = grr._cal_sym_recall(g["rect.sim"].to(dev), t2v, v2t)
    for k, v in got.items():
        assert abs(v - float(g["rect." + k])) < 1e-6, (k, v, float(g["rect." + k]))
    m = grr.GlobalRetrievalRecall(simi_logit_key=["l1_simi"])
    out = m.calculate(None, {"l1_simi": g["sq.sim"].to(dev)})
    assert abs(float(out["l1_simi_r@10"]) - float(g["sq.r@10"])) < 1e-6
    m.collect(None, {"l1_simi": g["rect.sim"][:9].to(dev)}, 0, 0, t2v=t2v[:9], v2t=v2t)
    m.collect(None, {"l1_simi": g["rect.sim"][9:].to(dev)}, 1, 0, t2v=t2v[9:])
    summ = m.summarize()
    assert abs(float(summ["l1_simi_t2v-mr"]) - float(g["rect.t2v-mr"])) < 1e-6 and abs(float(summ["l1_simi_v2t-r@5"]) - float(g["rect.v2t-r@5"])) < 1e-6
    # degenerate score matrices must not look like perfect retrieval: all-equal scores rank by column index (stable argsort of the
    # reference), a NaN ground-truth score ranks last
    flat = grr._cal_recall(torch.zeros(16, 16, device=dev))
    assert abs(flat["r@1"] - 1 / 16) < 1e-6 and abs(flat["r@5"] - 5 / 16) < 1e-6 and abs(flat["mr"] - 8.5) < 1e-6, flat
    nan = grr._cal_recall(torch.full((16, 16), float("nan"), device=dev))
    assert nan["r@1"] == 0.0 and nan["r@10"] == 0.0 and nan["mr"] == 16.0, nan
    ref_sorted = (-g["sq.sim"]).argsort(dim=1, stable=True)
    ref_rank = (ref_sorted == torch.arange(g["sq.sim"].shape[0])[:, None]).float().argmax(dim=1)
    assert torch.equal(grr.gt_ranks(g["sq.sim"].to(dev)).cpu().long(), ref_rank)


def case_m2_eval_retrieval(dev, golden, tmp_dir):
    """prj/M2_Encoder/eval_retrieval.py on the device path vs the reference's calu_recall / get_data run (metric_m2_recall.pt): the printed recalls
    (rounded to 0.1 as the reference prints them) and the ground-truth matrices built from a jsonl with curly quotes / upper case."""
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    m2 = os.path.join(root, "ant-multi-modal-framework_amd", "prj", "M2_Encoder")
    if m2 not in sys.path:
        sys.path.insert(0, m2)
    import eval_retrieval as er

    g = golden("metric_m2_recall.pt")
    out = er.calu_recall(g["txt"].to(dev), g["img"].to(dev), g["t2i_gt"], g["i2t_gt"], verbose=False)
    for k, want in zip((1, 5, 10), g["t2i_topk"].tolist()):
        assert abs(round(out[f"t2i_r@{k}"], 1) - want) < 0.051, (k, out, want)
    for k, want in zip((1, 5, 10), g["i2t_topk"].tolist()):
        assert abs(round(out[f"i2t_r@{k}"], 1) - want) < 0.051, (k, out, want)
    assert abs(round(out["MR"], 1) - float(g["MR"])) < 0.051
    path = os.path.join(tmp_dir, "pairs.jsonl")
    with open(path, "w") as f:
        f.write("\n".join(g["jsonl"]) + "\n")
    texts, images, t2i, i2t = er.get_data(path)
    assert texts == g["texts"] and images == g["images"]
    assert torch.equal(t2i, g["data.t2i_gt"]) and torch.equal(i2t, g["data.i2t_gt"])


# ------------------------------------------------------------------------------ dropout (counter-based masks)
def dropout_keep_np(idx, seed, p):
    """numpy twin of csrc/common.h::dropout_hash / DROPOUT_KEEP."""
    import numpy as np

    idx = np.asarray(idx, dtype=np.uint64)
    M = np.uint64(0xFFFFFFFF)
    h = (idx * np.uint64(0x9E3779B1) + np.uint64(seed & 0xFFFFFFFF)) & M
    h ^= h >> np.uint64(16); h = (h * np.uint64(0x85ebca6b)) & M
    h ^= h >> np.uint64(13); h = (h * np.uint64(0xc2b2ae35)) & M
    h ^= h >> np.uint64(16)
    h ^= np.uint64((seed >> 32) & 0xFFFFFFFF)
    h = (h * np.uint64(0x27d4eb2f)) & M
    h ^= h >> np.uint64(15)
    thr = np.uint64(0 if p <= 0 else int(float(np.float32(p)) * 4294967296.0))
    return torch.from_numpy((h >= thr))


def case_dropout(ops, dev):
    """dropout_add vs the numpy twin of the mask (exact), keep rate, residual path; attention with probability dropout vs the
    oracle attention with the same explicit mask (forward and all three gradients)."""
    import numpy as np

    n, p, seed = 8 * 1000, 0.25, (123456789 << 32) | 987654321
    x, r = rnd((n,), 501), rnd((n,), 502)
    keep = dropout_keep_np(np.arange(n), seed, p)
    assert abs(float(keep.float().mean()) - (1 - p)) < 0.02
    want = x * keep / (1 - p) + r
    check("dropout.add", ops.dropout_add(x.to(dev), p, seed, residual=r.to(dev)), want, 1e-6, 1e-6)
    check("dropout.bwd", ops.dropout_add(x.to(dev), p, seed), x * keep / (1 - p), 1e-6, 1e-6)
    check("dropout.bf16", ops.dropout_add(q(x).to(dev, BF), p, seed, residual=q(r).to(dev, BF)), q(x) * keep / (1 - p) + q(r), 2e-2, 2e-2)
    # attention
    B, heads, N, D = 2, 2, 21, 128
    scale, pa, seed = 0.125, 0.2, (77 << 32) | 4242
    qt, kt, vt, d_o = (q(rnd((B, N, D), 510 + i)) for i in range(4))
    lengths = torch.tensor([N, 9])
    key_bias = torch.zeros(B, N).masked_fill(~(torch.arange(N)[None, :] < lengths[:, None]), -10000.0)
    idx = np.arange(B * heads * N * N).reshape(B, heads, N, N)
    mask = dropout_keep_np(idx, seed, pa).float() / (1 - pa)
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (qt, kt, vt))
    s = torch.matmul(oops.split_heads(qr, heads), oops.split_heads(kr, heads).transpose(-1, -2)) * scale + key_bias[:, None, None, :]
    ref = oops.merge_heads(torch.matmul(torch.softmax(s, -1) * mask, oops.split_heads(vr, heads)))
    ref.backward(d_o)
    qd, kd, vd = qt.to(dev, BF), kt.to(dev, BF), vt.to(dev, BF)
    o, lse = ops.attention_fwd(qd, kd, vd, heads, scale, key_bias.to(dev), dropout_p=pa, dropout_seed=seed)
    check("attn.drop.o", o, ref, 2e-2, 2e-2)
    dq, dk, dv = ops.attention_bwd(qd, kd, vd, o, lse, d_o.to(dev, BF), heads, scale, key_bias.to(dev), dropout_p=pa, dropout_seed=seed)
    check("attn.drop.dq", dq, qr.grad, 3e-2, 3e-2)
    check("attn.drop.dk", dk, kr.grad, 3e-2, 3e-2)
    check("attn.drop.dv", dv, vr.grad, 3e-2, 3e-2)
    # key importance = head mean of the (dropped) probabilities summed over the queries (the reduction of `output_attentions` maps, univl_video_base.py:138-143),
    # accumulated into the caller's buffer: with the same dropout mask, and without dropout
    imp = torch.full((B, N), 0.5, device=dev)
    ops.attention_key_importance_(imp, qd, kd, lse, heads, scale, key_bias.to(dev), dropout_p=pa, dropout_seed=seed, weight=1.0 / heads)
    check("attn.drop.importance", imp, 0.5 + (torch.softmax(s, -1) * mask).detach().mean(1).sum(1), 2e-2, 2e-2)
    o0, lse0 = ops.attention_fwd(qd, kd, vd, heads, scale, key_bias.to(dev))
    imp0 = torch.zeros(B, N, device=dev)
    ops.attention_key_importance_(imp0, qd, kd, lse0, heads, scale, key_bias.to(dev), weight=1.0 / heads)
    check("attn.importance", imp0, torch.softmax(s, -1).detach().mean(1).sum(1), 2e-2, 2e-2)


def case_split_hi_lo(ops, dev):
    """ops.split_hi_lo against the torch form it replaces (cast, cast back, subtract, cast, zero pad): bit-identical, ragged shapes, a row-strided view."""
    odd = torch.randn(6, 1).expand(6, 1).as_strided((6, 1), (1, 6))     # one column with a non-unit inner stride (what a reshape can hand over)
    hi, lo = ops.split_hi_lo(odd.to(dev).as_strided((6, 1), (1, 6)))
    assert torch.equal(hi[:6, :1].cpu(), odd.to(BF)) and float(hi[:, 1:].float().abs().max()) == 0.0
    for rows, cols in ((5, 13), (64, 64), (1, 7), (33, 1024), (9, 1)):
        x = rnd((rows, cols), 71 + rows, 3.0)
        x[0, 0] = 1e-30; x[-1, -1] = -65504.0 * 3
        for view in (False, True):
            src = x
            if view:
                big = torch.zeros(rows, cols + 5)
                big[:, :cols] = x
                src = big.to(dev)[:, :cols]
            else:
                src = x.to(dev)
            hi, lo = ops.split_hi_lo(src)
            rp, cp = (rows + 7) // 8 * 8, (cols + 7) // 8 * 8
            assert hi.shape == (rp, cp) and lo.shape == (rp, cp)
            h_ref = x.to(BF)
            l_ref = (x - h_ref.float()).to(BF)
            assert torch.equal(hi[:rows, :cols].cpu(), h_ref) and torch.equal(lo[:rows, :cols].cpu(), l_ref), (rows, cols, view)
            pad = torch.cat([hi[rows:].flatten(), hi[:rows, cols:].flatten(), lo[rows:].flatten(), lo[:rows, cols:].flatten()])
            assert pad.numel() == 0 or float(pad.float().abs().max()) == 0.0
