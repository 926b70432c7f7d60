"""Tensor-level wrappers over the C ABI (no autograd here; see antmmf.hip.functional).

Every wrapper validates device / dtype / layout, passes raw pointers + torch's current HIP stream and
raises on a non-zero return code.  PyTorch is used for memory and streams only.
"""

import torch

from . import _lib
from ._lib import ACT_IDS, BF16, F32


def _stream():
    if _lib.backend() == 1:
        return torch.cuda.current_stream().cuda_stream
    return 0


def _dev_ok(*ts):
    want_cuda = _lib.backend() == 1
    for t in ts:
        if t is None:
            continue
        if t.is_cuda != want_cuda:
            raise RuntimeError(
                "antmmf.hip: tensors must live on the MI355X (cuda) for libantmmf_hip.so"
                if want_cuda else "antmmf.hip: the CPU lane emulator only accepts host tensors")


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"antmmf.hip: unsupported dtype {t.dtype}")


def _p(t):
    return 0 if t is None else t.data_ptr()


def _rc(rc, name):
    if rc != 0:
        raise RuntimeError(f"{name} failed with code {rc}")


def _c(t, name):
    if not t.is_contiguous():
        raise ValueError(f"antmmf.hip: {name} must be contiguous")
    return t


def _f32(t, name):
    if t is not None and t.dtype != torch.float32:
        raise TypeError(f"antmmf.hip: {name} must be float32")
    return t


# ------------------------------------------------------------------------------ LayerNorm
_LN_SCRATCH = {}


def layernorm_fwd(x, gamma, beta, eps, want_stats=True, act=None):
    """y = LN(act(x)); act=None is the plain LayerNorm."""
    _dev_ok(x, gamma, beta)
    _c(x, "x"); _f32(gamma, "gamma"); _f32(beta, "beta")
    cols = x.shape[-1]
    rows = x.numel() // cols
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if want_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if want_stats else None
    _rc(_lib.load().antmmf_act_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, cols, float(eps),
                                             ACT_IDS[act], _dt(x), _stream()), "antmmf_act_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, dgamma=None, dbeta=None, dres=None, act=None, dxsum=None):
    """Gradient of y = LN(act(x)) w.r.t. x (+ dres); dgamma / dbeta (fp32) are accumulated in place when given, and so is
    dxsum += column sums of the returned dx (the bias gradient of the Linear that produced x)."""
    _dev_ok(dy, x, mean, rstd, gamma, dgamma, dbeta, dres, dxsum)
    if dxsum is not None:
        _f32(dxsum, "dxsum")
    _c(dy, "dy"); _c(x, "x")
    if dres is not None:
        _c(dres, "dres")
    cols = x.shape[-1]
    rows = x.numel() // cols
    dx = torch.empty_like(x)
    scratch = None
    if dgamma is not None or dbeta is not None or dxsum is not None:  # per-workgroup column-sum partials (the library falls back to atomics without it)
        scratch = _LN_SCRATCH.get(x.device)
        if scratch is None or scratch.numel() < 1024 * 3 * cols:
            scratch = torch.empty(1024 * 3 * 4096, dtype=torch.float32, device=x.device)
            _LN_SCRATCH[x.device] = scratch
    _rc(_lib.load().antmmf_act_layernorm_bwd(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dres), _p(dx), _p(dgamma),
                                             _p(dbeta), _p(dxsum), rows, cols, ACT_IDS[act], _dt(x), _p(scratch),
                                             0 if scratch is None else scratch.numel(), _stream()), "antmmf_act_layernorm_bwd")
    return dx


def layernorm_bwd_renorm(dy, x, mean, rstd, gamma, beta, dgamma=None, dbeta=None, dres=None, dxsum=None):
    """Plain LayerNorm backward that also returns y = LN(x) again (one extra write instead of a recompute pass): -> (dx, y)."""
    _dev_ok(dy, x, mean, rstd, gamma, beta, dgamma, dbeta, dres, dxsum)
    _f32(gamma, "gamma"); _f32(beta, "beta")
    if dxsum is not None:
        _f32(dxsum, "dxsum")
    _c(dy, "dy"); _c(x, "x")
    if dres is not None:
        _c(dres, "dres")
    cols = x.shape[-1]
    rows = x.numel() // cols
    dx = torch.empty_like(x)
    y = torch.empty_like(x)
    _rc(_lib.load().antmmf_layernorm_bwd_renorm(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(dres), _p(dx), _p(y),
                                                _p(dgamma), _p(dbeta), _p(dxsum), rows, cols, _dt(x), _stream()), "antmmf_layernorm_bwd_renorm")
    return dx, y


# ------------------------------------------------------------------------------ activations
def act_fwd(u, act):
    _dev_ok(u); _c(u, "u")
    g = torch.empty_like(u)
    _rc(_lib.load().antmmf_act_fwd(_p(u), _p(g), u.numel(), ACT_IDS[act], _dt(u), _stream()), "antmmf_act_fwd")
    return g


def act_bwd(dg, u, act):
    _dev_ok(dg, u); _c(dg, "dg"); _c(u, "u")
    du = torch.empty_like(u)
    _rc(_lib.load().antmmf_act_bwd(_p(dg), _p(u), _p(du), u.numel(), ACT_IDS[act], _dt(u), _stream()), "antmmf_act_bwd")
    return du


# ------------------------------------------------------------------------------ L2 normalise
def l2norm_fwd(x, eps=1e-12, out_dtype=None):
    _dev_ok(x); _c(x, "x")
    out_dtype = out_dtype or x.dtype
    cols = x.shape[-1]
    rows = x.numel() // cols
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    inv = torch.empty(rows, dtype=torch.float32, device=x.device)
    _rc(_lib.load().antmmf_l2norm_fwd(_p(x), _p(y), _p(inv), rows, cols, float(eps), _dt(x), _dt(y), _stream()),
        "antmmf_l2norm_fwd")
    return y, inv


def l2norm_bwd(dy, y, inv, in_dtype):
    _dev_ok(dy, y, inv); _c(dy, "dy"); _c(y, "y")
    cols = y.shape[-1]
    rows = y.numel() // cols
    dx = torch.empty(y.shape, dtype=in_dtype, device=y.device)
    _rc(_lib.load().antmmf_l2norm_bwd(_p(dy), _p(y), _p(inv), _p(dx), rows, cols, _dt(dx), _dt(y), _stream()),
        "antmmf_l2norm_bwd")
    return dx


# ------------------------------------------------------------------------------ small movers
def colsum_(out, x2d):
    """out[c] += sum_r x2d[r, c]; x2d may be a row-strided 2-D view (stride(1) == 1)."""
    _dev_ok(out, x2d); _f32(out, "out")
    if x2d.dim() != 2 or x2d.stride(1) != 1:
        raise ValueError("colsum_: x2d must be 2-D with unit inner stride")
    _rc(_lib.load().antmmf_colsum(_p(x2d), _p(out), x2d.shape[0], x2d.shape[1], x2d.stride(0), _dt(x2d), _stream()),
        "antmmf_colsum")
    return out


def transpose_bf16(x):
    _dev_ok(x); _c(x, "x")
    assert x.dim() == 2 and x.dtype == torch.bfloat16
    out = torch.empty(x.shape[1], x.shape[0], dtype=x.dtype, device=x.device)
    _rc(_lib.load().antmmf_transpose_bf16(_p(x), _p(out), x.shape[0], x.shape[1], _stream()), "antmmf_transpose_bf16")
    return out


def transpose_bf16_batched(in_base, out_base, table, n_mats, total_tiles):
    """out_base[out_off : out_off + rows * cols] = in_base[in_off : ...].view(rows, cols).t() for every row of `table` (see antmmf_hip.h)."""
    _dev_ok(in_base, out_base, table)
    assert in_base.dtype == torch.bfloat16 and out_base.dtype == torch.bfloat16 and table.dtype == torch.int64 and table.is_contiguous()
    _rc(_lib.load().antmmf_transpose_bf16_batched(_p(in_base), _p(out_base), _p(table), int(n_mats), int(total_tiles), _stream()),
        "antmmf_transpose_bf16_batched")


def cast_bf16(x, out=None):
    _dev_ok(x, out); _c(x, "x"); _f32(x, "x")
    if out is None:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    _rc(_lib.load().antmmf_cast_f32_bf16(_p(x), _p(out), x.numel(), _stream()), "antmmf_cast_f32_bf16")
    return out


def split_hi_lo(x, r_mult=8, c_mult=8):
    """fp32 [rows, cols] (unit inner stride) -> (hi, lo) bf16, zero-padded to multiples of (r_mult, c_mult): hi = bf16(x), lo = bf16(x - hi).  One launch."""
    _dev_ok(x); _f32(x, "x")
    if x.dim() != 2 or (x.shape[1] > 1 and x.stride(1) != 1):      # (a single column may carry any inner stride)
        raise ValueError("split_hi_lo: x must be 2-D with unit inner stride")
    rows, cols = x.shape
    rp, cp = (rows + r_mult - 1) // r_mult * r_mult, (cols + c_mult - 1) // c_mult * c_mult
    if cp % 8:
        raise ValueError("split_hi_lo: the padded column count must be a multiple of 8")
    hi = torch.empty(rp, cp, dtype=torch.bfloat16, device=x.device)
    lo = torch.empty(rp, cp, dtype=torch.bfloat16, device=x.device)
    _rc(_lib.load().antmmf_split_hi_lo_bf16(_p(x), x.stride(0) if rows > 1 else max(cols, 1), rows, cols, _p(hi), _p(lo), rp, cp, _stream()), "antmmf_split_hi_lo_bf16")
    return hi, lo


def patchify(img, patch, kpad=None, shift=0.0, scale=1.0):
    _dev_ok(img); _c(img, "img")
    b, c, h, w = img.shape
    kk = c * patch * patch
    kpad = kpad or ((kk + 63) // 64) * 64
    out = torch.empty(b * (h // patch) * (w // patch), kpad, dtype=torch.bfloat16, device=img.device)
    _rc(_lib.load().antmmf_patchify(_p(img), _p(out), b, c, h, w, patch, kpad, float(shift), float(scale), _dt(img),
                                    _stream()), "antmmf_patchify")
    return out


def assemble_tokens(patch_tokens, cls, pos, bias, batch, grid):
    _dev_ok(patch_tokens, cls, pos, bias); _c(patch_tokens, "patch_tokens")
    _f32(cls, "cls"); _f32(pos, "pos"); _f32(bias, "bias")
    d = patch_tokens.shape[-1]
    out = torch.empty(batch, grid + 1, d, dtype=torch.bfloat16, device=patch_tokens.device)
    _rc(_lib.load().antmmf_assemble_tokens(_p(patch_tokens), _p(cls), _p(pos), _p(bias), _p(out), batch, grid, d,
                                           _stream()), "antmmf_assemble_tokens")
    return out


def split_tokens(dx):
    _dev_ok(dx); _c(dx, "dx")
    b, n, d = dx.shape
    out = torch.empty(b * (n - 1), d, dtype=torch.bfloat16, device=dx.device)
    _rc(_lib.load().antmmf_split_tokens(_p(dx), _p(out), b, n - 1, d, _stream()), "antmmf_split_tokens")
    return out


def embed_gather(ids, word, pos=None, type_table=None, type_ids=None, zero_rows=None, pos_offset=0):
    _dev_ok(ids, word, pos, type_table, type_ids, zero_rows); _c(ids, "ids")
    assert ids.dtype == torch.int64
    b, seq = ids.shape
    d = word.shape[1]
    out = torch.empty(b, seq, d, dtype=torch.bfloat16, device=word.device)
    if zero_rows is not None:
        assert zero_rows.dtype == torch.uint8 and zero_rows.is_contiguous()
    _rc(_lib.load().antmmf_embed_gather(_p(ids), _p(word), _p(pos), _p(type_table), _p(type_ids), _p(zero_rows), _p(out),
                                        b * seq, seq, d, pos_offset, _stream()), "antmmf_embed_gather")
    return out


SCATTER_SORT_MIN_ROWS = 8192   # below this the element-wise atomic kernel is cheaper than a device sort


def embed_scatter_add_(dtable, dx, idx=None, skip_rows=None, seq=1, offset=0):
    _dev_ok(dtable, dx, idx, skip_rows); _c(dx, "dx"); _f32(dtable, "dtable")
    d = dx.shape[-1]
    rows = dx.numel() // d
    if idx is not None and rows >= SCATTER_SORT_MIN_ROWS and dtable.is_contiguous():
        # table rows with one writer each instead of one fp32 atomic per element: sort the ids (torch's device radix sort: plumbing), then one wave per run
        keys = idx.reshape(-1)
        if skip_rows is not None:
            keys = torch.where(skip_rows.reshape(-1) != 0, torch.full_like(keys, dtable.shape[0]), keys)
        sorted_idx, src_row = torch.sort(keys, stable=True)   # stable: the rows of a run are summed in batch order -- deterministic, unlike the atomics
        _rc(_lib.load().antmmf_embed_scatter_add_sorted(_p(dx), _p(sorted_idx), _p(src_row), _p(dtable), rows, dtable.shape[0], d, _stream()),
            "antmmf_embed_scatter_add_sorted")
        return dtable
    _rc(_lib.load().antmmf_embed_scatter_add(_p(dx), _p(idx), _p(skip_rows), _p(dtable), rows, seq, d, offset, _stream()),
        "antmmf_embed_scatter_add")
    return dtable


def adamw_step_(p, g, m, v, shadow, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    """grad_scale: a Python number, or a 1-element fp32 DEVICE tensor (e.g. 1/world x clip coefficient) that is read by the kernel."""
    _dev_ok(p, g, m, v, shadow)
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _c(t, n); _f32(t, n)
    if torch.is_tensor(grad_scale):
        gs = grad_scale.detach().reshape(-1)[:1].float().contiguous()
        _dev_ok(p, gs)
        _rc(_lib.load().antmmf_adamw_step_scaled(_p(p), _p(g), _p(m), _p(v), _p(shadow), p.numel(), float(lr), float(beta1),
                                                 float(beta2), float(eps), float(weight_decay), int(step), 1.0, _p(gs),
                                                 _stream()), "antmmf_adamw_step_scaled")
        return
    _rc(_lib.load().antmmf_adamw_step(_p(p), _p(g), _p(m), _p(v), _p(shadow), p.numel(), float(lr), float(beta1),
                                      float(beta2), float(eps), float(weight_decay), int(step), float(grad_scale),
                                      _stream()), "antmmf_adamw_step")


def sumsq_(out, x):
    _dev_ok(out, x); _c(x, "x"); _f32(x, "x"); _f32(out, "out")
    _rc(_lib.load().antmmf_sumsq(_p(x), _p(out), x.numel(), _stream()), "antmmf_sumsq")
    return out


# ------------------------------------------------------------------------------ GEMM
# bench.py sets GEMM_TRACE to a list: every launch is then bracketed by HIP events on the launch stream and
# (start, end, flops, layout-tag) is appended -- the live per-launch timing behind bench.py's `roofline` object.
GEMM_TRACE = None




def gemm(P, Q, out=None, p_rmajor=False, q_rmajor=False, out_dtype=torch.bfloat16, alpha=1.0, bias=None, act=None,
         residual=None, aux=None, gate=None, accumulate=False, split_k=1, aux_grad=False, gate_is_grad=False):
    """out[i, j] = epi(alpha * sum_r P[i, r] Q[j, r]).  P is [I, R] (or [R, I] when p_rmajor), Q likewise.
    2-D operands with unit inner stride; row strides are passed through (views of packed buffers are fine).
    aux_grad: `aux` receives act'(pre-activation) instead of the pre-activation; gate_is_grad: `gate` holds act' already."""
    _dev_ok(P, Q, out, bias, residual, aux, gate)
    for t, n in ((P, "P"), (Q, "Q")):
        if t.dim() != 2 or t.stride(1) != 1 or t.dtype != torch.bfloat16:
            raise ValueError(f"gemm: {n} must be a 2-D bf16 tensor with unit inner stride")
    I, R = (P.shape[1], P.shape[0]) if p_rmajor else (P.shape[0], P.shape[1])
    J, R2 = (Q.shape[1], Q.shape[0]) if q_rmajor else (Q.shape[0], Q.shape[1])
    if R != R2:
        raise ValueError(f"gemm: reduction mismatch {R} vs {R2}")
    if out is None:
        out = torch.empty(I, J, dtype=out_dtype, device=P.device)
    if out.dim() != 2 or out.stride(1) != 1 or tuple(out.shape) != (I, J):
        raise ValueError("gemm: bad output")
    _f32(bias, "bias")

    def ld(t):
        return 0 if t is None else t.stride(0)

    for t, n in ((residual, "residual"), (aux, "aux"), (gate, "gate")):
        if t is not None and (t.dtype != torch.bfloat16 or t.dim() != 2 or t.stride(1) != 1 or tuple(t.shape) != (I, J)):
            raise ValueError(f"gemm: bad {n}")
    ev = None
    if GEMM_TRACE is not None and _lib.backend() == 1:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    # (no scratch: since round 5 the persistent NT kernel finishes the leftover tiles of its walk as cells inside the same launch -- csrc/gemm.hip, gemm_nt_k64r_kernel)
    ws = None
    _rc(_lib.load().antmmf_gemm_bf16_ws(_p(P), _p(Q), _p(out), I, J, R, P.stride(0), Q.stride(0), out.stride(0),
                                        int(p_rmajor), int(q_rmajor), _dt(out), float(alpha), _p(bias),
                                        ACT_IDS[act] | (0x100 if aux_grad else 0) | (0x200 if gate_is_grad else 0),
                                        _p(residual), ld(residual), _p(aux), ld(aux), _p(gate), ld(gate), int(accumulate),
                                        int(split_k), _p(ws), 0 if ws is None else ws.numel() * 4, _stream()), "antmmf_gemm_bf16_ws")
    if ev is not None:
        ev[1].record()
        GEMM_TRACE.append((ev[0], ev[1], 2.0 * I * J * R, ("tn" if p_rmajor else ("nn" if q_rmajor else "nt")),
                           (I, J, R, ("b" if bias is not None else "") + ("r" if residual is not None else "") + (act or "") + ("g" if gate is not None else ""))))
    return out


def gemm_gated_colsum_ok(I, J, R, ldc=None, ldgate=None):
    """True when gemm_gated_colsum serves the shape (the rolling-epilogue kernel's shapes: I, J multiples of 256, J <= 4096, R % 64 == 0, R >= 192, >= 512 tiles)."""
    return bool(_lib.load().antmmf_gemm_bf16_gated_colsum_ok(int(I), int(J), int(R), int(J if ldc is None else ldc), int(J if ldgate is None else ldgate)))


def gemm_gated_colsum(P, Q, gate, act=None):
    """(out, parts): out[i, j] = (sum_r P[i, r] Q[j, r]) * gate[i, j]  (gate = the activation derivative the forward stored) and parts [256, J] fp32 whose column sums
    are the column sums of out -- the bias gradient of the Linear in front of the activation without a pass over the [I, J] tensor.  P [I, R], Q [J, R] bf16."""
    _dev_ok(P, Q, gate)
    for t, n in ((P, "P"), (Q, "Q"), (gate, "gate")):
        if t.dim() != 2 or t.stride(1) != 1 or t.dtype != torch.bfloat16:
            raise ValueError(f"gemm_gated_colsum: {n} must be a 2-D bf16 tensor with unit inner stride")
    I, R = P.shape
    J = Q.shape[0]
    if Q.shape[1] != R or tuple(gate.shape) != (I, J):
        raise ValueError("gemm_gated_colsum: shape mismatch")
    out = torch.empty(I, J, dtype=torch.bfloat16, device=P.device)
    parts = torch.zeros(256, J, dtype=torch.float32, device=P.device)
    ev = None
    if GEMM_TRACE is not None and _lib.backend() == 1:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    _rc(_lib.load().antmmf_gemm_bf16_gated_colsum(_p(P), _p(Q), _p(out), I, J, R, P.stride(0), Q.stride(0), out.stride(0), _p(gate), gate.stride(0), _p(parts), _stream()),
        "antmmf_gemm_bf16_gated_colsum")
    if ev is not None:
        ev[1].record()
        GEMM_TRACE.append((ev[0], ev[1], 2.0 * I * J * R, "nt", (I, J, R, (act or "") + "g")))
    return out, parts


_WGRAD_WS = {}


def gemm_wgrad_(dW, dY, X, split_k_hint=1):
    """dW[n_out, k_in] += dY[tokens, n_out]^T X[tokens, k_in]   (fp32 accumulate in place; bf16 token-major operands,
    row-strided views allowed).  A per-device fp32 workspace for the token-split partial sums is kept and reused."""
    _dev_ok(dW, dY, X); _f32(dW, "dW")
    for t, n in ((dY, "dY"), (X, "X")):
        if t.dim() != 2 or t.stride(1) != 1 or t.dtype != torch.bfloat16:
            raise ValueError(f"gemm_wgrad_: {n} must be a 2-D bf16 tensor with unit inner stride")
    tokens, n_out = dY.shape
    k_in = X.shape[1]
    if X.shape[0] != tokens or tuple(dW.shape) != (n_out, k_in) or dW.stride(1) != 1:
        raise ValueError("gemm_wgrad_: shape mismatch")
    need = 32 * n_out * k_in
    key = (dW.device, _stream())      # per (device, stream), as for the NT scratch
    ws = _WGRAD_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1 << 24), dtype=torch.float32, device=dW.device)
        _WGRAD_WS[key] = ws
    ev = None
    if GEMM_TRACE is not None and _lib.backend() == 1:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    _rc(_lib.load().antmmf_gemm_wgrad_bf16(_p(dY), _p(X), _p(dW), tokens, n_out, k_in, dY.stride(0), X.stride(0), dW.stride(0),
                                           int(split_k_hint), _p(ws), ws.numel() * 4, _stream()), "antmmf_gemm_wgrad_bf16")
    if ev is not None:
        ev[1].record()
        GEMM_TRACE.append((ev[0], ev[1], 2.0 * tokens * n_out * k_in, "tn", (n_out, k_in, tokens, "wgrad")))
    return dW


def gemm_wgrad_seg_(dWs, dY, X, split_k_hint=1):
    """dWs[s][seg_rows, k_in] += dY[:, s * seg_rows : (s + 1) * seg_rows]^T X for the <= 4 equally shaped fp32 buffers in `dWs` (unrelated addresses: the separate
    q / k / v weights of a layer in the gradient arena) as ONE wgrad GEMM over the packed dY (csrc/gemm.hip antmmf_gemm_wgrad_bf16_seg)."""
    import ctypes

    _dev_ok(dY, X, *dWs)
    for t, n in ((dY, "dY"), (X, "X")):
        if t.dim() != 2 or t.stride(1) != 1 or t.dtype != torch.bfloat16:
            raise ValueError(f"gemm_wgrad_seg_: {n} must be a 2-D bf16 tensor with unit inner stride")
    n_seg = len(dWs)
    seg_rows, k_in = dWs[0].shape
    tokens = dY.shape[0]
    if not 1 <= n_seg <= 4 or dY.shape[1] != n_seg * seg_rows or X.shape != (tokens, k_in):
        raise ValueError("gemm_wgrad_seg_: shape mismatch")
    ld = dWs[0].stride(0)
    for w in dWs:
        _f32(w, "dW")
        if tuple(w.shape) != (seg_rows, k_in) or w.stride(1) != 1 or w.stride(0) != ld:
            raise ValueError("gemm_wgrad_seg_: the segments must have one shape and one row stride")
    need = 32 * n_seg * seg_rows * k_in
    key = (dY.device, _stream())
    ws = _WGRAD_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1 << 24), dtype=torch.float32, device=dY.device)
        _WGRAD_WS[key] = ws
    ptrs = (ctypes.c_void_p * n_seg)(*[w.data_ptr() for w in dWs])
    ev = None
    if GEMM_TRACE is not None and _lib.backend() == 1:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    _rc(_lib.load().antmmf_gemm_wgrad_bf16_seg(_p(dY), _p(X), ctypes.cast(ptrs, ctypes.c_void_p), n_seg, seg_rows, tokens, k_in, dY.stride(0), X.stride(0), ld,
                                               int(split_k_hint), _p(ws), ws.numel() * 4, _stream()), "antmmf_gemm_wgrad_bf16_seg")
    if ev is not None:
        ev[1].record()
        GEMM_TRACE.append((ev[0], ev[1], 2.0 * tokens * n_seg * seg_rows * k_in, "tn", (n_seg * seg_rows, k_in, tokens, "wgrad")))
    return dWs


# ------------------------------------------------------------------------------ M2 feed-forward with the sub-LayerNorm folded into its GEMMs
_FFN_WS = {}


def _ffn_ws(dev, nbytes):
    ws = _FFN_WS.get(dev)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty(max((nbytes + 3) // 4, 1 << 22), dtype=torch.float32, device=dev)
        _FFN_WS[dev] = ws
    return ws


def _bf2d(t, name):
    if t.dim() != 2 or t.stride(1) != 1 or t.dtype != torch.bfloat16:
        raise ValueError(f"ffn: {name} must be a 2-D bf16 tensor with unit inner stride")
    return t


def _trace_begin():
    if GEMM_TRACE is not None and _lib.backend() == 1:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        return ev
    return None


def _trace_end(ev, I, J, R, tag):
    if ev is not None:
        ev[1].record()
        GEMM_TRACE.append((ev[0], ev[1], 2.0 * I * J * R, "nt", (I, J, R, tag)))


def ffn_prepare_w2(W2, gamma, beta, b2):
    """fp32 master W2 [n_out, n_ff], ffn_layernorm (gamma, beta), fc2 bias -> (W2g bf16 = W2 diag(gamma), c = row sums of the rounded W2g, b2f = b2 + W2 beta)."""
    _dev_ok(W2, gamma, beta, b2); _f32(W2, "W2"); _f32(gamma, "gamma"); _f32(beta, "beta"); _f32(b2, "b2"); _c(W2, "W2")
    n_out, n_ff = W2.shape
    w2g = torch.empty(n_out, n_ff, dtype=torch.bfloat16, device=W2.device)
    c = torch.empty(n_out, dtype=torch.float32, device=W2.device)
    b2f = torch.empty(n_out, dtype=torch.float32, device=W2.device)
    _rc(_lib.load().antmmf_ffn_prepare_w2(_p(W2), _p(gamma), _p(beta), _p(b2), _p(w2g), _p(c), _p(b2f), n_out, n_ff, _stream()), "antmmf_ffn_prepare_w2")
    return w2g, c, b2f


def ffn_fc1_fwd(x, W1, b1, act, eps):
    """-> (z = act(x W1^T + b1), dact = act'(.), stats [tokens, 2] = (mean, rstd) of the rounded rows of z)."""
    _dev_ok(x, W1, b1); _bf2d(x, "x"); _bf2d(W1, "W1"); _f32(b1, "b1")
    tokens, n_in = x.shape
    n_ff = W1.shape[0]
    z = torch.empty(tokens, n_ff, dtype=torch.bfloat16, device=x.device)
    dact = torch.empty_like(z)
    stats = torch.empty(tokens, 2, dtype=torch.float32, device=x.device)
    ws = _ffn_ws(x.device, (n_ff // 64) * tokens * 8)
    ev = _trace_begin()
    _rc(_lib.load().antmmf_ffn_fc1_fwd(_p(x), _p(W1), _p(b1), _p(z), _p(dact), _p(stats), tokens, n_ff, n_in, x.stride(0), W1.stride(0), z.stride(0),
                                       ACT_IDS[act], float(eps), _p(ws), ws.numel() * 4, _stream()), "antmmf_ffn_fc1_fwd")
    _trace_end(ev, tokens, n_ff, n_in, "ffn1")
    return z, dact, stats


def ffn_fc2_fwd(z, w2g, c, b2f, stats, res):
    _dev_ok(z, w2g, c, b2f, stats, res); _bf2d(z, "z"); _bf2d(w2g, "w2g"); _bf2d(res, "res"); _f32(c, "c"); _f32(b2f, "b2f"); _f32(stats, "stats")
    tokens, n_ff = z.shape
    n_out = w2g.shape[0]
    y = torch.empty(tokens, n_out, dtype=torch.bfloat16, device=z.device)
    ev = _trace_begin()
    _rc(_lib.load().antmmf_ffn_fc2_fwd(_p(z), _p(w2g), _p(c), _p(b2f), _p(stats), _p(res), _p(y), tokens, n_out, n_ff, z.stride(0), w2g.stride(0), res.stride(0),
                                       y.stride(0), _stream()), "antmmf_ffn_fc2_fwd")
    _trace_end(ev, tokens, n_out, n_ff, "ffn2")
    return y


def ffn_bwd_rows(dy, y, res, b2f, c, stats, n_ff, s_col, cs_col=None):
    """-> (rowv4 [tokens, 4] = (mu, rstd, m1, m2), dys = bf16(rstd dy)); accumulates s_col (and cs_col when given)."""
    _dev_ok(dy, y, res, b2f, c, stats, s_col, cs_col); _bf2d(dy, "dy"); _bf2d(y, "y"); _bf2d(res, "res"); _f32(s_col, "s_col"); _f32(cs_col, "cs_col")
    tokens, n_out = dy.shape
    rowv4 = torch.empty(tokens, 4, dtype=torch.float32, device=dy.device)
    dys = torch.empty(tokens, n_out, dtype=torch.bfloat16, device=dy.device)
    _rc(_lib.load().antmmf_ffn_bwd_rows(_p(dy), _p(y), _p(res), _p(b2f), _p(c), _p(stats), _p(rowv4), _p(dys), _p(s_col), _p(cs_col), tokens, n_out, int(n_ff),
                                        dy.stride(0), y.stride(0), res.stride(0), dys.stride(0), _stream()), "antmmf_ffn_bwd_rows")
    return rowv4, dys


def ffn_fc2_dgrad(dy, w2gt, z, dact, rowv4, db1=None):
    """du = act'(u) * LayerNorm-backward(dy W2g) -- the gradient at fc1's pre-activation; db1 += column sums of du."""
    _dev_ok(dy, w2gt, z, dact, rowv4, db1); _bf2d(dy, "dy"); _bf2d(w2gt, "w2gt"); _bf2d(z, "z"); _bf2d(dact, "dact"); _f32(db1, "db1")
    tokens, n_out = dy.shape
    n_ff = w2gt.shape[0]
    if z.stride(0) != dact.stride(0):
        raise ValueError("ffn_fc2_dgrad: z and dact must share their row stride")
    du = torch.empty(tokens, n_ff, dtype=torch.bfloat16, device=dy.device)
    ws = _ffn_ws(dy.device, (tokens // 128 + 1) * n_ff * 4)
    ev = _trace_begin()
    _rc(_lib.load().antmmf_ffn_fc2_dgrad(_p(dy), _p(w2gt), _p(z), _p(dact), _p(rowv4), _p(du), _p(db1), tokens, n_ff, n_out, dy.stride(0), w2gt.stride(0),
                                         z.stride(0), du.stride(0), _p(ws), ws.numel() * 4, _stream()), "antmmf_ffn_fc2_dgrad")
    _trace_end(ev, tokens, n_ff, n_out, "ffn3")
    return du


def ffn_wgrad_post_(dW2, Gm, W2, gamma, beta, s_col, cs_col, dgamma=None, dbeta=None):
    _dev_ok(dW2, Gm, W2, gamma, beta, s_col, cs_col, dgamma, dbeta)
    for t, n in ((dW2, "dW2"), (Gm, "Gm"), (W2, "W2")):
        _f32(t, n); _c(t, n)
    n_out, n_ff = W2.shape
    _rc(_lib.load().antmmf_ffn_wgrad_post(_p(Gm), _p(W2), _p(gamma), _p(beta), _p(s_col), _p(cs_col), _p(dW2), _p(dgamma), _p(dbeta), n_out, n_ff, _stream()),
        "antmmf_ffn_wgrad_post")
    return dW2


# ------------------------------------------------------------------------------ attention
def _tok_ld(t, name):
    if t.dim() != 3 or t.stride(2) != 1 or t.stride(0) != t.shape[1] * t.stride(1) or t.dtype != torch.bfloat16:
        raise ValueError(f"attention: {name} must be a [B, N, heads*64] bf16 view with unit inner stride and dense batch stride")
    return t.stride(1)


def _head_dim(D, heads):
    if D % heads or D // heads not in (64, 128):
        raise ValueError(f"attention: head size {D}/{heads} is not supported by the HIP kernels (64 or 128)")
    return D // heads


def attention_fwd(q, k, v, heads, scale, key_bias=None, dropout_p=0.0, dropout_seed=0):
    """dropout_p > 0: attention-probability dropout with the counter-based mask of (dropout_seed, element index).  Head size =
    q.shape[-1] / heads: 64 or 128."""
    _dev_ok(q, k, v, key_bias)
    B, Nq, D = q.shape
    Nk = k.shape[1]
    dh = _head_dim(D, heads)
    o = torch.empty(B, Nq, D, dtype=torch.bfloat16, device=q.device)
    lse = torch.empty(B, heads, Nq, dtype=torch.float32, device=q.device)
    if key_bias is not None:
        _f32(key_bias, "key_bias"); _c(key_bias, "key_bias")
        assert tuple(key_bias.shape) == (B, Nk)
    _rc(_lib.load().antmmf_attention_fwd_hd(_p(q), _p(k), _p(v), _p(key_bias), _p(o), _p(lse), B, heads, dh, Nq, Nk,
                                            _tok_ld(q, "q"), _tok_ld(k, "k"), _tok_ld(v, "v"), D, float(scale), float(dropout_p),
                                            int(dropout_seed), _stream()),
        "antmmf_attention_fwd_hd")
    return o, lse


def attention_key_importance_(out, q, k, lse, heads, scale, key_bias=None, dropout_p=0.0, dropout_seed=0, weight=1.0):
    """out [B, Nk] fp32 += weight * sum over heads and queries of the attention probabilities of (q, k) -- see antmmf_attention_key_importance."""
    _dev_ok(out, q, k, lse, key_bias); _f32(out, "out"); _c(out, "out"); _c(lse, "lse")
    B, Nq, D = q.shape
    Nk = k.shape[1]
    assert _head_dim(D, heads) == 64 and tuple(out.shape) == (B, Nk)
    _rc(_lib.load().antmmf_attention_key_importance(_p(q), _p(k), _p(key_bias), _p(lse), _p(out), B, heads, Nq, Nk, _tok_ld(q, "q"), _tok_ld(k, "k"), float(scale),
                                                    float(dropout_p), int(dropout_seed), float(weight), _stream()), "antmmf_attention_key_importance")
    return out


def attention_bwd_sums_ok(head_dim, Nq, Nk, dropout_p=0.0):
    """True when attention_bwd(..., sums=...) is served: the one-kernel backward's shapes (head size 64, no dropout, 33 ... 272 keys)."""
    return bool(_lib.load().antmmf_attention_bwd_sums_ok(int(head_dim), int(Nq), int(Nk), float(dropout_p)))


def attention_bwd(q, k, v, o, lse, d_o, heads, scale, key_bias=None, dq=None, dk=None, dv=None, dropout_p=0.0, dropout_seed=0, sums=None, sums_v=True):
    """`sums` [B, 3 * D] fp32 (contiguous): also filled with the per-batch-item token sums of dQ | dK | dV -- the q / k / v bias gradients are its column sums
    (check attention_bwd_sums_ok first: only the one-kernel backward's shapes are served); sums_v=False leaves the dV third unwritten."""
    _dev_ok(q, k, v, o, lse, d_o, key_bias, dq, dk, dv, sums)
    B, Nq, D = q.shape
    Nk = k.shape[1]
    dh = _head_dim(D, heads)
    _c(o, "o"); _c(d_o, "d_o")
    if dq is None:
        dq = torch.empty(B, Nq, D, dtype=torch.bfloat16, device=q.device)
    if dk is None:
        dk = torch.empty(B, Nk, D, dtype=torch.bfloat16, device=q.device)
    if dv is None:
        dv = torch.empty(B, Nk, D, dtype=torch.bfloat16, device=q.device)
    if sums is not None:
        _c(sums, "sums"); _f32(sums, "sums")
        if tuple(sums.shape) != (B, 3 * D) or dh != 64 or dropout_p:
            raise ValueError("attention_bwd: sums must be [B, 3 * D] fp32 with head size 64 and no dropout")
        _rc(_lib.load().antmmf_attention_bwd_sums(_p(q), _p(k), _p(v), _p(key_bias), _p(o), _p(lse), _p(d_o), _p(dq), _p(dk), _p(dv), _p(sums),
                                                  1 if sums_v else 0, B, heads, Nq, Nk, _tok_ld(q, "q"), _tok_ld(k, "k"), _tok_ld(v, "v"), D, D,
                                                  _tok_ld(dq, "dq"), _tok_ld(dk, "dk"), _tok_ld(dv, "dv"), float(scale), _stream()),
            "antmmf_attention_bwd_sums")
        return dq, dk, dv
    _rc(_lib.load().antmmf_attention_bwd_hd(_p(q), _p(k), _p(v), _p(key_bias), _p(o), _p(lse), _p(d_o), _p(dq), _p(dk), _p(dv),
                                            B, heads, dh, Nq, Nk, _tok_ld(q, "q"), _tok_ld(k, "k"), _tok_ld(v, "v"), D, D,
                                            _tok_ld(dq, "dq"), _tok_ld(dk, "dk"), _tok_ld(dv, "dv"), float(scale), float(dropout_p),
                                            int(dropout_seed), _stream()),
        "antmmf_attention_bwd_hd")
    return dq, dk, dv


# ------------------------------------------------------------------------------ losses
def milnce_fwd(Rm, Cm, n_pair, row_offset):
    _dev_ok(Rm, Cm); _c(Rm, "Rm"); _c(Cm, "Cm"); _f32(Rm, "Rm"); _f32(Cm, "Cm")
    B = Rm.shape[0]
    loss_rows = torch.empty(B, dtype=torch.float32, device=Rm.device)
    denom = torch.empty(B, dtype=torch.float32, device=Rm.device)
    _rc(_lib.load().antmmf_milnce_fwd(_p(Rm), _p(Cm), B, Rm.shape[1], Cm.shape[1], n_pair, row_offset, _p(loss_rows),
                                      _p(denom), _stream()), "antmmf_milnce_fwd")
    return loss_rows, denom


def milnce_bwd(Rm, Cm, denom, coef, n_pair, row_offset, out_dtype=torch.bfloat16):
    _dev_ok(Rm, Cm, denom, coef); _f32(coef, "coef"); _c(coef, "coef")
    dR = torch.empty(Rm.shape, dtype=out_dtype, device=Rm.device)
    dC = torch.empty(Cm.shape, dtype=out_dtype, device=Rm.device)
    _rc(_lib.load().antmmf_milnce_bwd(_p(Rm), _p(Cm), _p(denom), _p(coef), Rm.shape[0], Rm.shape[1], Cm.shape[1], n_pair,
                                      row_offset, _p(dR), _p(dC), _dt(dR), _stream()), "antmmf_milnce_bwd")
    return dR, dC


def softmax_ce_fwd(x, row_offset, log_scale=None, scale_mul=1.0):
    _dev_ok(x, log_scale); _c(x, "x"); _f32(x, "x"); _f32(log_scale, "log_scale")
    B, Wd = x.shape
    loss_rows = torch.empty(B, dtype=torch.float32, device=x.device)
    lse = torch.empty(B, dtype=torch.float32, device=x.device)
    _rc(_lib.load().antmmf_softmax_ce_fwd(_p(x), B, Wd, row_offset, _p(log_scale), float(scale_mul), _p(loss_rows), _p(lse),
                                          _stream()), "antmmf_softmax_ce_fwd")
    return loss_rows, lse


def softmax_ce_bwd(x, lse, coef, row_offset, log_scale=None, scale_mul=1.0, dscale=None, out_dtype=torch.bfloat16):
    _dev_ok(x, lse, coef, log_scale, dscale); _f32(coef, "coef"); _c(coef, "coef")
    dx = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    _rc(_lib.load().antmmf_softmax_ce_bwd(_p(x), _p(lse), _p(coef), x.shape[0], x.shape[1], row_offset, _p(log_scale),
                                          float(scale_mul), _p(dx), _p(dscale), _dt(dx), _stream()), "antmmf_softmax_ce_bwd")
    return dx


def moco_fwd(pos, neg, temperature):
    """Rows of MocoUtils.moco_loss: LSE([pos, neg] / T) - LSE(pos / T); pos [R, Np], neg [R, K] fp32."""
    _dev_ok(pos, neg); _c(pos, "pos"); _c(neg, "neg"); _f32(pos, "pos"); _f32(neg, "neg")
    R = pos.shape[0]
    loss_rows = torch.empty(R, dtype=torch.float32, device=pos.device)
    lse_all, lse_pos = torch.empty_like(loss_rows), torch.empty_like(loss_rows)
    _rc(_lib.load().antmmf_moco_fwd(_p(pos), _p(neg), R, pos.shape[1], neg.shape[1], 1.0 / float(temperature), _p(loss_rows), _p(lse_all),
                                    _p(lse_pos), _stream()), "antmmf_moco_fwd")
    return loss_rows, lse_all, lse_pos


def moco_bwd(pos, neg, lse_all, lse_pos, coef, temperature, out_dtype=torch.bfloat16):
    _dev_ok(pos, neg, lse_all, lse_pos, coef); _f32(coef, "coef"); _c(coef, "coef")
    dpos = torch.empty_like(pos)
    dneg = torch.empty(neg.shape, dtype=out_dtype, device=neg.device)
    _rc(_lib.load().antmmf_moco_bwd(_p(pos), _p(neg), _p(lse_all), _p(lse_pos), _p(coef), pos.shape[0], pos.shape[1], neg.shape[1],
                                    1.0 / float(temperature), _p(dpos), _p(dneg), _dt(dneg), _stream()), "antmmf_moco_bwd")
    return dpos, dneg


def ema_update_(k, q, m, k_shadow=None):
    """k = m k + (1 - m) q in place over flat fp32 ranges (same numel); k_shadow (bf16, optional) is rewritten with bf16(k)."""
    _dev_ok(k, q, k_shadow); _f32(k, "k"); _f32(q, "q"); _c(k, "k"); _c(q, "q")
    if k.numel() != q.numel():
        raise ValueError("ema_update_: size mismatch")
    if k_shadow is not None and (k_shadow.dtype != torch.bfloat16 or k_shadow.numel() != k.numel() or not k_shadow.is_contiguous()):
        raise ValueError("ema_update_: k_shadow must be a contiguous bf16 tensor of the same size")
    _rc(_lib.load().antmmf_ema_update(_p(k), _p(q), _p(k_shadow), k.numel(), float(m), _stream()), "antmmf_ema_update")
    return k


def negnce_fwd(S, diag, row_offset=0, scale=100.0, margin=0.0):
    """Per-row NegNCE pieces on a row slab S [B, W]: (-log p_ii, sum of -log(1 - p_ij) over violating negatives, their count, LSE)."""
    _dev_ok(S, diag); _c(S, "S"); _c(diag, "diag"); _f32(S, "S"); _f32(diag, "diag")
    B, W = S.shape
    outs = [torch.empty(B, dtype=torch.float32, device=S.device) for _ in range(4)]
    _rc(_lib.load().antmmf_negnce_fwd(_p(S), _p(diag), B, W, row_offset, float(scale), float(margin), *[_p(o) for o in outs], _stream()),
        "antmmf_negnce_fwd")
    return tuple(outs)


def negnce_bwd(S, diag, lse, coef, row_offset=0, scale=100.0, margin=0.0, out_dtype=torch.float32):
    """coef: 2-float device tensor (weight of every positive row term, weight of every selected negative term)."""
    _dev_ok(S, diag, lse, coef); _f32(coef, "coef"); _c(coef, "coef")
    dS = torch.empty(S.shape, dtype=out_dtype, device=S.device)
    _rc(_lib.load().antmmf_negnce_bwd(_p(S), _p(diag), _p(lse), _p(coef), S.shape[0], S.shape[1], row_offset, float(scale), float(margin),
                                      _p(dS), _dt(dS), _stream()), "antmmf_negnce_bwd")
    return dS


def wti_reduce_fwd(S, A, T, B, V, tmask, vmask, f2f=None, z2_of=None):
    """S [A*T, B*V] fp32 -> (t2v [A,B,T], v2t [A,B,V], z1 [A,B,T] int32, tmax [A,B,V] int32); see include/antmmf_hip.h."""
    _dev_ok(S, tmask, vmask, f2f, z2_of); _c(S, "S"); _f32(S, "S")
    dev = S.device
    t2v = torch.empty(A, B, T, dtype=torch.float32, device=dev)
    v2t = torch.empty(A, B, V, dtype=torch.float32, device=dev)
    z1 = torch.empty(A, B, T, dtype=torch.int32, device=dev)
    tmax = torch.empty(A, B, V, dtype=torch.int32, device=dev)
    _rc(_lib.load().antmmf_wti_reduce_fwd(_p(S), A, T, B, V, _p(tmask), _p(vmask), _p(f2f), _p(z2_of), _p(t2v), _p(v2t), _p(z1), _p(tmax),
                                          _stream()), "antmmf_wti_reduce_fwd")
    return t2v, v2t, z1, tmax


def wti_reduce_bwd(S, A, T, B, V, tmask, vmask, f2f, z2_of, z1, tmax, dt2v, dv2t, out_dtype=torch.bfloat16):
    _dev_ok(S, tmask, vmask, f2f, z2_of, z1, tmax, dt2v, dv2t)
    dt2v, dv2t = dt2v.float().contiguous(), dv2t.float().contiguous()  # bound to locals: the launch reads them after this line
    dS = torch.empty(S.shape, dtype=out_dtype, device=S.device)
    df2f = torch.zeros(B, V, dtype=torch.float32, device=S.device) if f2f is not None else None
    _rc(_lib.load().antmmf_wti_reduce_bwd(_p(S), A, T, B, V, _p(tmask), _p(vmask), _p(f2f), _p(z2_of), _p(z1), _p(tmax), _p(dt2v),
                                          _p(dv2t), _p(dS), _p(df2f), _dt(dS), _stream()), "antmmf_wti_reduce_bwd")
    return dS, df2f


# ------------------------------------------------------------------------------ DMAE stage-3 head: token weights, aligned-pair products
_TW_SCRATCH = {}


def _f32c(t, name):
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise TypeError(f"antmmf.hip: {name} must be a contiguous float32 tensor")
    return t


def token_weight_fwd(feat, w, bias, mask):
    """softmax over the tokens of (feat . w + bias), masked tokens (mask < 0.5) excluded: feat [N, T, D] fp32 -> [N, T]."""
    _dev_ok(feat, w, bias, mask); _f32c(feat, "feat"); _f32c(w, "w")
    N, T, D = feat.shape
    if mask is not None:
        _f32c(mask, "mask")
    out = torch.empty(N, T, dtype=torch.float32, device=feat.device)
    _rc(_lib.load().antmmf_token_weight_fwd(_p(feat), _p(w), _p(bias), _p(mask), _p(out), N, T, D, _stream()), "antmmf_token_weight_fwd")
    return out


def token_weight_bwd(feat, w, p, dout, dw, dbias=None, want_dfeat=True):
    """-> dfeat (or None); dw [D] / dbias [1] are accumulated in place."""
    _dev_ok(feat, w, p, dout, dw, dbias)
    for t, n in ((feat, "feat"), (w, "w"), (p, "p"), (dout, "dout"), (dw, "dw")):
        _f32c(t, n)
    N, T, D = feat.shape
    scratch = _TW_SCRATCH.get(feat.device)
    if scratch is None or scratch.numel() < 512 * (D + 1):
        scratch = torch.empty(512 * 1025, dtype=torch.float32, device=feat.device)
        _TW_SCRATCH[feat.device] = scratch
    dfeat = torch.empty_like(feat) if want_dfeat else None
    _rc(_lib.load().antmmf_token_weight_bwd(_p(feat), _p(w), _p(p), _p(dout), _p(dfeat), _p(dw), _p(dbias), _p(scratch), scratch.numel(), N, T, D,
                                            _stream()), "antmmf_token_weight_bwd")
    return dfeat


def pair_dots(x, y):
    """out[c, v] = x[c, :] . y[c, v, :]   (x [C, D], y [C, V, D] fp32)."""
    _dev_ok(x, y); _f32c(x, "x"); _f32c(y, "y")
    C, V, D = y.shape
    if tuple(x.shape) != (C, D):
        raise ValueError("pair_dots: x must be [C, D]")
    out = torch.empty(C, V, dtype=torch.float32, device=y.device)
    _rc(_lib.load().antmmf_pair_dots(_p(x), _p(y), _p(out), C, V, D, _stream()), "antmmf_pair_dots")
    return out


def pair_wsum(w, y):
    """out[c, :] = sum_v w[c, v] y[c, v, :]."""
    _dev_ok(w, y); _f32c(w, "w"); _f32c(y, "y")
    C, V, D = y.shape
    if tuple(w.shape) != (C, V):
        raise ValueError("pair_wsum: w must be [C, V]")
    out = torch.empty(C, D, dtype=torch.float32, device=y.device)
    _rc(_lib.load().antmmf_pair_wsum(_p(w), _p(y), _p(out), C, V, D, _stream()), "antmmf_pair_wsum")
    return out


def pair_outer(w, x):
    """out[c, v, :] = w[c, v] x[c, :]."""
    _dev_ok(w, x); _f32c(w, "w"); _f32c(x, "x")
    C, V = w.shape
    D = x.shape[1]
    out = torch.empty(C, V, D, dtype=torch.float32, device=x.device)
    _rc(_lib.load().antmmf_pair_outer(_p(w), _p(x), _p(out), C, V, D, _stream()), "antmmf_pair_outer")
    return out


def tis_keep(w, thresh):
    """keep mask [R, T] of TokenImportanceSelector: 0 for the tokens whose descending cumulative weight (inclusive) is < thresh."""
    _dev_ok(w); _f32c(w, "w")
    R, T = w.shape
    keep = torch.empty_like(w)
    _rc(_lib.load().antmmf_tis_keep(_p(w), float(thresh), _p(keep), R, T, _stream()), "antmmf_tis_keep")
    return keep


def rank_rows(S, gt_off, gt_idx):
    """rank [rows] int32: position (0 = first) of the best ground-truth column of every row of S [rows, cols] fp32;
    gt_off [rows + 1] / gt_idx int32 list the ground-truth columns of each row (CSR)."""
    _dev_ok(S, gt_off, gt_idx); _f32(S, "S")
    if S.dim() != 2 or S.stride(1) != 1:
        raise ValueError("rank_rows: S must be 2-D with unit inner stride")
    rank = torch.empty(S.shape[0], dtype=torch.int32, device=S.device)
    _rc(_lib.load().antmmf_rank_rows(_p(S), S.stride(0), S.shape[0], S.shape[1], _p(gt_off), _p(gt_idx), _p(rank), _stream()), "antmmf_rank_rows")
    return rank


def dropout_add(x, p, seed, residual=None):
    """y = x * keep / (1 - p) (+ residual) with the counter-based mask keep(seed, element index); its own backward w.r.t. x is the
    same call on dy without residual."""
    _dev_ok(x, residual); _c(x, "x")
    if residual is not None:
        _c(residual, "residual")
    y = torch.empty_like(x)
    _rc(_lib.load().antmmf_dropout_add(_p(x), _p(residual), _p(y), x.numel(), float(p), int(seed), _dt(x), _stream()), "antmmf_dropout_add")
    return y
