"""Autograd functions that put the HIP kernels behind torch's tape.

Design (MI355X-first, sized for 288 GB HBM at per-GPU batch 1024):
  * activations are bf16, parameters are fp32 masters with a bf16 compute shadow (antmmf.hip.arena);
  * one autograd node per transformer LAYER (`transformer_layer`): it saves only x, the packed QKV,
    the attention context + log-sum-exp, the mid-layer residual stream and the MLP pre-activation
    (20 bytes per token-channel) and RECOMPUTES the cheap HBM-bound pieces (LayerNorm outputs,
    GELU outputs) in backward -- the reference keeps every intermediate alive (~34 B / token-channel);
  * weight gradients are accumulated by the wgrad GEMM straight into the fp32 gradient arena when the
    parameter lives in one (the function then returns None for that input), otherwise returned.

Reference modules replaced by `transformer_layer`:
  "clip"  ResidualAttentionBlock   antmmf/modules/vision/backbone/clip/model.py:227-256
  "bert"  BertLayer                antmmf/modules/vision/backbone/clip/modeling_bert.py:134-270
  "m2"    torchscale EncoderLayer  prj/M2_Encoder/vlmo/torchscale/architecture/encoder.py:113-168
"""
import math
from dataclasses import dataclass

import os

import torch

from . import _lib, ops

BF = torch.bfloat16


# ------------------------------------------------------------------------------ parameter helpers
def compute_copy(p):
    """bf16 compute copy of an fp32 master parameter (arena shadow if present, else cast now)."""
    if p is None:
        return None
    s = getattr(p, "_antmmf_bf16", None)
    if s is not None:
        # the shadow is rewritten by the fused optimizer / EMA launches (which bypass torch's version counter); any OTHER in-place write
        # to the fp32 master -- load_state_dict, an initialiser, p.clamp_() -- bumps p._version and is picked up here.  (Writes through
        # `p.data` are invisible to the counter: call arena.sync_shadow() after those.)
        if getattr(p, "_antmmf_ver", None) != p._version:
            ops.cast_bf16(p.detach().reshape(-1), out=s.view(-1))
            p._antmmf_ver = p._version
            bump_weight_version()
        return s
    if p.dtype == BF:
        return p.detach()
    return ops.cast_bf16(p.detach().contiguous())


def f32(p):
    return None if p is None else p.detach()


_T_CACHE_VERSION = [0]


def bump_weight_version():
    """Called by the optimizer after every parameter update: invalidates the cached transposed weight copies."""
    _T_CACHE_VERSION[0] += 1


def compute_copy_t(p):
    """bf16 TRANSPOSE of a 2-D master weight ([out, in] -> [in, out]); lets dgrad (dX = dY W) run the all-r-contiguous
    LDS-DMA GEMM.  Cached on the parameter until the next optimizer step (weights are tiny next to activations).  Arena parameters that
    asked for it once are re-transposed by the optimizer step in ONE launch (refresh_transposes) instead of one launch each here."""
    if getattr(p, "_antmmf_bf16", None) is not None and getattr(p, "_antmmf_ver", None) != p._version:
        compute_copy(p)          # an out-of-band write to the master: refreshes the shadow and bumps the weight version
    ver = _T_CACHE_VERSION[0]
    c = getattr(p, "_antmmf_bf16_t", None)
    if c is not None and c[0] == ver and getattr(p, "_antmmf_main_grad", None) is not None:
        return c[1]
    t = ops.transpose_bf16(compute_copy(p).contiguous())
    try:
        p._antmmf_bf16_t = (ver, t)
        arena = getattr(p, "_antmmf_arena", None)
        if arena is not None and p.dim() == 2:
            reg = arena.__dict__.setdefault("_t_registry", {})
            reg.setdefault(id(p), p)
    except AttributeError:
        pass
    return t


def refresh_transposes(arena):
    """After an optimizer step (the bf16 shadow is fresh, the weight version just bumped): every transposed copy that the previous step's
    backward asked for, in one launch out of the arena's shadow into one persistent buffer."""
    reg = getattr(arena, "_t_registry", None)
    if not reg:
        return
    plan = getattr(arena, "_t_plan", None)
    if plan is None or plan["count"] != len(reg):
        params = list(reg.values())
        rows, tiles, out_off = [], 0, 0
        for p in params:
            r, c = p.shape
            rows.append([p._antmmf_offset, out_off, r, c, tiles])
            tiles += ((r + 63) // 64) * ((c + 63) // 64)
            out_off += (r * c + 7) // 8 * 8
        dev = arena.shadow.device
        plan = dict(count=len(reg), params=params, tiles=tiles, table=torch.tensor(rows, dtype=torch.int64, device=dev).contiguous(),
                    out=torch.empty(out_off, dtype=BF, device=dev), offs=[r[1] for r in rows])
        arena._t_plan = plan
    ver = _T_CACHE_VERSION[0]
    stale = [p for p in plan["params"] if getattr(p, "_antmmf_ver", None) != p._version]
    for p in stale:          # a parameter written outside the optimizer since: its shadow is refreshed by compute_copy, per parameter
        compute_copy(p)
    ver = _T_CACHE_VERSION[0]
    ops.transpose_bf16_batched(arena.shadow, plan["out"], plan["table"], plan["count"], plan["tiles"])
    for p, off in zip(plan["params"], plan["offs"]):
        r, c = p.shape
        p._antmmf_bf16_t = (ver, plan["out"][off:off + r * c].view(c, r))


def dgrad(dy2d, weight, weight_layout="oi", **epi):
    """dX = dY W for W stored [out, in] ("oi") or dX = dY W^T for W stored [in, out] ("io")."""
    if weight_layout == "io":
        return ops.gemm(dy2d, compute_copy(weight), **epi)          # Q = W [in][out]: already r-contiguous
    if torch.is_tensor(weight) and isinstance(weight, torch.nn.Parameter):
        return ops.gemm(dy2d, compute_copy_t(weight), **epi)        # Q = W^T [in][out]
    return ops.gemm(dy2d, weight if weight.dtype == BF else compute_copy(weight), q_rmajor=True, **epi)  # ad-hoc tensor (e.g. concatenated qkv)


class GradSink:
    """Where weight gradients go: the parameter's slice of the fp32 gradient arena (accumulated in
    place, autograd gets None) or a fresh fp32 tensor that is handed back to autograd."""

    def __init__(self):
        self._fresh = {}

    def buf(self, p):
        mg = getattr(p, "_antmmf_main_grad", None)
        if mg is not None:
            return mg
        if id(p) not in self._fresh:
            self._fresh[id(p)] = torch.zeros(p.shape, dtype=torch.float32, device=p.device)
        return self._fresh[id(p)]

    def result(self, p, needs):
        if p is None or not needs:
            return None
        if getattr(p, "_antmmf_main_grad", None) is not None:
            return None
        return self._fresh.get(id(p))


def _wgrad(sink, W, dy2d, x2d, w_is_in_out=False):
    """dW += dy^T x  (W stored [out, in]);  for W stored [in, out] (CLIP `proj`) dW += x^T dy."""
    if W is None or not W.requires_grad:
        return
    out = sink.buf(W)
    tiles = ((out.shape[0] + 127) // 128) * ((out.shape[1] + 127) // 128)
    tokens = dy2d.shape[0]
    # token split for the kernels that take the hint (the small / unaligned shapes; the BK = 64 wgrad kernel sizes its own): enough workgroups to cover the chip even
    # when the weight is a handful of 128 x 128 tiles -- a [32, 16] predictor weight over 98304 pair-tokens ran as ONE workgroup x 8 splits for 2.1 ms per call
    # (6.3 ms of the dmae12 step), the 768-wide text tower at 3840 tokens as 36 workgroups without any split
    split = 1
    if tiles < 256 and tokens >= 4096:
        split = min(64, max(1, 512 // tiles), tokens // 2048)
    elif tiles <= 64 and tokens >= 1024:     # (measured: with more tiles the atomic accumulation of the splits costs more than the idle CUs, 768 x 3072 at 3840 tokens 1.09 -> 1.40 ms)
        split = min(8, max(1, 256 // tiles), tokens // 512)
    if w_is_in_out:
        ops.gemm_wgrad_(out, x2d, dy2d, split)
    else:
        ops.gemm_wgrad_(out, dy2d, x2d, split)


def _bgrad(sink, b, dy2d):
    if b is None or not b.requires_grad:
        return
    ops.colsum_(sink.buf(b), dy2d)


def _note_untracked(*params):
    """Generic ops write parameter gradients through GradSink without telling the arena when they are final: while an overlapped gradient
    all-reduce is armed, their parameters' buckets must wait for the end of the backward pass (arena.note_untracked)."""
    if not torch.is_grad_enabled():
        return
    for p in params:
        arena = getattr(p, "_antmmf_arena", None) if p is not None else None
        if arena is not None and getattr(arena, "_ov", None) is not None:
            arena.note_untracked(params)
            return


# ------------------------------------------------------------------------------ generic ops
class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        x = x.contiguous()
        y, mean, rstd = ops.layernorm_fwd(x, f32(weight), f32(bias), eps)
        ctx.save_for_backward(x, mean, rstd, weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, weight, bias = ctx.saved_tensors
        sink = GradSink()
        dg = sink.buf(weight) if weight.requires_grad else None
        db = sink.buf(bias) if bias.requires_grad else None
        dx = ops.layernorm_bwd(dy.contiguous(), x, mean, rstd, f32(weight), dg, db)
        return dx, sink.result(weight, ctx.needs_input_grad[1]), sink.result(bias, ctx.needs_input_grad[2]), None


def layer_norm(x, weight, bias, eps):
    _note_untracked(weight, bias)
    return _LayerNorm.apply(x, weight, bias, eps)


class _Linear(torch.autograd.Function):
    """y = act(x W^T + b) (+ residual).  weight_layout "oi": W is [out, in] (nn.Linear); "io": W is [in, out]
    (CLIP `proj` / `text_projection`, used as x @ W).
    out_f32 (round 6; for small heads whose state should not be rounded to bf16 between layers -- the opt-in fp32 stream of DMAE's temporal transformer): x may be fp32
    (rounded to bf16 only as the GEMM's operand), y and dx are fp32 (the MFMA's fp32 accumulators stored unrounded)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, residual, weight_layout, out_f32=False):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        ctx.x_f32 = x2.dtype == torch.float32
        if ctx.x_f32:
            x2 = ops.cast_bf16(x2)
        W = compute_copy(weight)
        io = weight_layout == "io"
        n_out = W.shape[1] if io else W.shape[0]
        if out_f32 and (act or residual is not None):
            raise ValueError("linear(out_f32=True): plain bias epilogue only")
        aux = torch.empty(x2.shape[0], n_out, dtype=BF, device=x.device) if act else None
        res2 = residual.reshape(-1, n_out) if residual is not None else None
        y = ops.gemm(x2, W, q_rmajor=io, bias=f32(bias), act=act, residual=res2, aux=aux, out_dtype=torch.float32 if out_f32 else BF)
        ctx.save_for_backward(x2, weight, bias, aux)
        ctx.act, ctx.io, ctx.shp, ctx.has_res = act, io, shp, residual is not None
        return y.view(*shp[:-1], n_out)

    @staticmethod
    def backward(ctx, dy):
        x2, weight, bias, aux = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        sink = GradSink()
        if dy2.dtype == torch.float32:   # out_f32: the bias gradient from the unrounded gradient, the GEMM operand rounded once
            _bgrad(sink, bias, dy2)
            dy2 = ops.cast_bf16(dy2)
            du = dy2
        else:
            du = ops.act_bwd(dy2, aux, ctx.act) if ctx.act else dy2
            _bgrad(sink, bias, du)
        _wgrad(sink, weight, du, x2, w_is_in_out=ctx.io)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = dgrad(du, weight, "io" if ctx.io else "oi", **(dict(out_dtype=torch.float32) if ctx.x_f32 else {})).view(ctx.shp)
        return (dx, sink.result(weight, ctx.needs_input_grad[1]), sink.result(bias, bias is not None and ctx.needs_input_grad[2]),
                None, dy if ctx.has_res else None, None, None)


def linear(x, weight, bias=None, act=None, residual=None, weight_layout="oi", out_f32=False):
    _note_untracked(weight, bias)
    return _Linear.apply(x, weight, bias, act, residual, weight_layout, out_f32)


class _L2Norm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps):
        x = x.contiguous()
        y, inv = ops.l2norm_fwd(x, eps, out_dtype=torch.float32)
        ctx.save_for_backward(y, inv)
        ctx.in_dtype = x.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        return ops.l2norm_bwd(dy.contiguous().float(), y, inv, ctx.in_dtype), None


def l2_normalize(x, eps=1e-12):
    """bf16 (or fp32) rows -> fp32 unit rows (the embeddings feed the fp32-accurate similarity path)."""
    return _L2Norm.apply(x, eps)


class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, key_bias, heads, scale):
        o, lse = ops.attention_fwd(q, k, v, heads, scale, key_bias)
        ctx.save_for_backward(q, k, v, o, lse, key_bias)
        ctx.heads, ctx.scale = heads, scale
        return o

    @staticmethod
    def backward(ctx, d_o):
        q, k, v, o, lse, key_bias = ctx.saved_tensors
        dq, dk, dv = ops.attention_bwd(q, k, v, o, lse, d_o.contiguous(), ctx.heads, ctx.scale, key_bias)
        return dq, dk, dv, None, None, None


def attention(q, k, v, heads, scale, key_bias=None):
    """softmax(scale q k^T + key_bias) v on [B, N, heads*64] bf16 tensors (views of a packed projection are fine)."""
    return _Attention.apply(q, k, v, key_bias, heads, scale)


def co_attention(q1, k1, v1, q2, k2, v2, heads, bias1=None, bias2=None):
    """ViLBERT BertBiAttention core (antmmf/models/vilbert.py:360-400): stream-2 queries attend stream-1
    keys/values and vice versa; two launches of the same kernel with the streams swapped."""
    scale = 1.0 / math.sqrt(64)
    return attention(q2, k1, v1, heads, scale, bias1), attention(q1, k2, v2, heads, scale, bias2)


# ------------------------------------------------------------------------------ fused transformer layer
@dataclass(frozen=True)
class LayerSpec:
    kind: str          # "clip" | "bert" | "m2"
    heads: int
    eps: float
    act: str           # "quick_gelu" | "gelu"
    packed_qkv: bool   # True: one [3d, d] in_proj (CLIP);  False: separate q/k/v weights
    attn_dropout: float = 0.0    # BERT only, training only: attention_probs_dropout_prob (modeling_bert.py:157)
    hidden_dropout: float = 0.0  # BERT only, training only: hidden_dropout_prob after both dense layers (:175-186,227-238)


# parameter slots of `transformer_layer` (absent ones are None)
SLOTS = ("ln1_w", "ln1_b", "wqkv", "bqkv", "wq", "bq", "wk", "bk", "wv", "bv", "inner_w", "inner_b", "wo", "bo",
         "ln2_w", "ln2_b", "w1", "b1", "ffn_w", "ffn_b", "w2", "b2")


def _cached_on(p, attr, build):
    """Per-optimizer-step cache hung on an arena-managed parameter (same validity rule as compute_copy_t)."""
    ver = _T_CACHE_VERSION[0]
    c = getattr(p, attr, None)
    if c is not None and c[0] == ver and getattr(p, "_antmmf_main_grad", None) is not None:
        return c[1]
    val = build()
    try:
        setattr(p, attr, (ver, val))
    except AttributeError:
        pass
    return val


def _arena_run(params):
    """(arena, first offset) when `params` sit back to back in ONE arena, in this order (arena.tag_pack asked for it and their sizes are multiples of the arena's
    alignment) -- their concatenation is then a view of the arena's buffers; None otherwise."""
    arena = getattr(params[0], "_antmmf_arena", None) if params[0] is not None else None
    if arena is None:
        return None
    off = params[0]._antmmf_offset
    nxt = off
    for p in params:
        if p is None or getattr(p, "_antmmf_arena", None) is not arena or p._antmmf_offset != nxt:
            return None
        nxt += p.numel()
    return arena, off


def _packed_qkv_weight(P, spec):
    if spec.packed_qkv:
        return compute_copy(P["wqkv"]), f32(P["bqkv"])
    ws, bs = [P["wq"], P["wk"], P["wv"]], [P["bq"], P["bk"], P["bv"]]
    rw, rb = _arena_run(ws), _arena_run(bs)
    if rw is not None and rb is not None:
        # q / k / v weights adjacent in the arena (round 6): the [3d, d] operand is a view of the bf16 shadow, the [3d] bias a view of the fp32 master -- no copy at all
        # (before: torch.cat of three weights, three biases and three transposed weights per layer and optimizer step = 7 launches x 48 layers of the flagship)
        for w in ws:
            compute_copy(w)      # (out-of-band writes to a master are picked up per parameter here; no launch when the shadow is fresh)
        d = ws[0].shape[1]
        return rw[0].shadow[rw[1]:rw[1] + 3 * ws[0].numel()].view(3 * ws[0].shape[0], d), rb[0].master[rb[1]:rb[1] + 3 * bs[0].numel()]
    # separate q / k / v projections (BERT, torchscale): one [3d, d] GEMM operand, rebuilt once per optimizer step
    w = _cached_on(P["wq"], "_antmmf_qkv_w", lambda: torch.cat([compute_copy(P["wq"]), compute_copy(P["wk"]), compute_copy(P["wv"])], dim=0))
    b = _cached_on(P["wq"], "_antmmf_qkv_b", lambda: torch.cat([f32(P["bq"]), f32(P["bk"]), f32(P["bv"])], dim=0))
    return w, b


def _packed_qkv_weight_t(P, spec):
    if spec.packed_qkv:
        return compute_copy_t(P["wqkv"])                                   # [d, 3d]
    if _arena_run([P["wq"], P["wk"], P["wv"]]) is not None:
        # the transpose of the packed view, once per optimizer step (one launch; before: three transposes + a cat along the columns)
        return _cached_on(P["wq"], "_antmmf_qkv_wt", lambda: ops.transpose_bf16(_packed_qkv_weight(P, spec)[0]))
    return _cached_on(P["wq"], "_antmmf_qkv_wt",
                      lambda: torch.cat([compute_copy_t(P["wq"]), compute_copy_t(P["wk"]), compute_copy_t(P["wv"])], dim=1))


# Activation-memory policy.  By default a layer saves 20 B per token-channel (x, qkv, o, mid, u); the normalised tensors come back out of
# the LayerNorm backward kernels (layernorm_bwd_renorm).  KEEP_FFN_NORM additionally keeps fc2's input g_n = ffn_layernorm(gelu(u)) (M2) or
# act(u) (CLIP / BERT layers) (+8 B per token-channel, the widest recompute: one full pass over the 4d-wide tensor per layer) -- worth it
# when HBM allows (288 GB on MI355X: the ViT-L/14 step at 1024 pairs/GPU peaks at ~175 GiB without it).  Set through set_keep_ffn_norm().
KEEP_FFN_NORM = False
COLSUM_HANDOFFS = [0]   # diagnostic: fc2 bias gradients taken from the next layer's ln1 backward instead of a column-sum pass
DEBUG_NO_HANDOFF = False  # debugging aid (a Python attribute, not an environment variable): always run the column-sum pass


def set_keep_ffn_norm(flag):
    global KEEP_FFN_NORM
    KEEP_FFN_NORM = bool(flag)


# Sub-LN fold (M2 layers under the KEEP_FFN_NORM activation policy): gelu -> ffn_layernorm live in the epilogues of fc1 / fc2 and of fc2's dgrad
# (ops.ffn_*, csrc/gemm.hip "Sub-LN fold"); the two 4d-wide LayerNorm passes per layer disappear.  Same activation memory as the kept policy
# (z = gelu(u) and gelu'(u) instead of u and LN(z)).  OFF by default: measured on MI355X at the bench size (tools/ffn_fold_bench.py,
# profiles/r3_ffn_fold_bench_image.jsonl) the removed passes (1.05 + 1.43 ms per image layer) come back one for one as exposed epilogue time of the
# persistent GEMM kernel (fc1 + 0.92, dgrad + 0.96, fc2 + 0.11, row pass 0.46): its HBM-bound epilogues do not overlap the MFMA loop, so moving bytes
# from a streaming kernel into them buys nothing (step 1314 vs 1313 pairs/s).  Since round 5 its entry points live in the LAB library only
# (libantmmf_hip_lab.so, include/antmmf_hip_lab.h): set_ffn_fold(True) works when that library is the loaded one (tests, tools/ffn_fold_bench.py) and raises otherwise.
FFN_FOLD = False
# q / k / v bias gradients out of the attention backward's per-item token sums (round 6; module flag for the A/B, tools/colsum_ab.py)
ATTN_BWD_SUMS = True
# q / k / v wgrads as one segmented launch on the long towers too when the merged launch fills at least this many of the 256 CUs (d = 768: 27 tiles x 9 splits = 243; d = 1024: 48 x 5 = 240, measured a loss)
QKV_WGRAD_MERGE_MIN_WGS = 243
QK_WGRAD_MERGE = True   # (d = 1024, long tower: q | k as one launch of 256 workgroups, v alone)
# fc1's bias gradient of the CLIP / BERT feed-forwards out of the gated dgrad's epilogue (round 6; module flag for the A/B)
GATED_DGRAD_COLSUM = True
ATTN_BWD_SUMS_MIN_TOKENS = 33    # (tools/bench_flag.py A/B: 129 = the long towers only)
# Keep the output of the pre-LN layers' second LayerNorm (fc1's input, + 2 B per token-channel: 15.7 GiB on the l14 step at 1024 pairs) for backward instead of having the
# LayerNorm backward re-emit it (5 -> 4 tensor streams in that kernel).  EXPERIMENT (round 6, tools/bench_flag.py KEEP_LN2_OUT=1): off; see docs/rounds/round-6.md
KEEP_LN2_OUT = False
if os.environ.get("ANTMMF_FFN_FOLD"):   # rounds 3 - 4 read this variable; since round 5 the fold is an experiment of the lab library behind set_ffn_fold()
    import warnings

    warnings.warn("ANTMMF_FFN_FOLD is set but ignored: the sub-LN fold is a lab-library experiment (antmmf.hip.functional.set_ffn_fold(True) with "
                  "ANTMMF_HIP_LIB=.../libantmmf_hip_lab.so); the product path always runs the two 4d-wide LayerNorm passes", stacklevel=2)


# Collector for the one thing the reference does with attention MAPS on this path (univl_video_base.py:131-143: words_importance = sum over layers of the
# head-mean attention, summed over the queries): a [B, N] fp32 tensor while a BertEncoder runs with output_attentions=True, None otherwise.
KEY_IMPORTANCE = None


def set_ffn_fold(flag):
    global FFN_FOLD
    if flag and not _lib.is_lab():
        raise RuntimeError("the sub-LN fold is an experiment of the lab library (make -C csrc lab; ANTMMF_HIP_LIB=.../libantmmf_hip_lab.so): the product library does not export it")
    FFN_FOLD = bool(flag)


def _folded_w2(P):
    """(W2 diag(gamma) bf16, its transpose, c = its row sums, b2f = b2 + W2 beta) -- rebuilt once per optimizer step (two small launches per layer)."""
    def build():
        w2g, c, b2f = ops.ffn_prepare_w2(f32(P["w2"]), f32(P["ffn_w"]), f32(P["ffn_b"]), f32(P["b2"]))
        return w2g, ops.transpose_bf16(w2g), c, b2f

    return _cached_on(P["w2"], "_antmmf_ffn_fold", build)


class _TransformerLayer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, key_bias, spec, seed, *params):
        P = dict(zip(SLOTS, params))
        p_att, p_hid = (spec.attn_dropout, spec.hidden_dropout) if spec.kind == "bert" else (0.0, 0.0)
        seed = int(seed or 0)
        B, N, d = x.shape
        x2 = x.reshape(B * N, d)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        T = B * N
        dev = x.device
        scale = 64 ** -0.5
        wqkv, bqkv = _packed_qkv_weight(P, spec)
        pre_ln = spec.kind in ("clip", "m2")
        st1 = None
        if pre_ln:
            h, m1, r1 = ops.layernorm_fwd(x2, f32(P["ln1_w"]), f32(P["ln1_b"]), spec.eps)
            st1 = (m1, r1)
        else:
            h = x2
        qkv = ops.gemm(h, wqkv, bias=bqkv)  # [T, 3d]
        del h
        q3 = qkv.view(B, N, 3 * d)
        o, lse = ops.attention_fwd(q3[..., :d], q3[..., d:2 * d], q3[..., 2 * d:], spec.heads, scale, key_bias, p_att, seed)
        if KEY_IMPORTANCE is not None:   # `output_attentions=True` callers: the head-mean column sums of this layer's probabilities join the collector
            ops.attention_key_importance_(KEY_IMPORTANCE, q3[..., :d], q3[..., d:2 * d], lse, spec.heads, scale, key_bias, p_att, seed or 0, weight=1.0 / spec.heads)
        o2 = o.view(T, d)
        st_in = None
        if spec.kind == "m2":
            o_n, mi, ri = ops.layernorm_fwd(o2, f32(P["inner_w"]), f32(P["inner_b"]), spec.eps)
            st_in = (mi, ri)
        else:
            o_n = o2
        if p_hid > 0:  # LayerNorm(dropout(dense(ctx)) + x): the dropout sits between the bias and the residual add
            x1 = ops.dropout_add(ops.gemm(o_n, compute_copy(P["wo"]), bias=f32(P["bo"])), p_hid, seed + 1, residual=x2)
        else:
            x1 = ops.gemm(o_n, compute_copy(P["wo"]), bias=f32(P["bo"]), residual=x2)  # attention output + residual
        del o_n
        if pre_ln:
            mid = x1  # residual stream after attention
            h2, m2_, r2 = ops.layernorm_fwd(mid, f32(P["ln2_w"]), f32(P["ln2_b"]), spec.eps)
            st2 = (m2_, r2)
        else:  # bert: a = LN(s1)
            mid = x1  # s1 (pre-LN sum)
            h2, m2_, r2 = ops.layernorm_fwd(mid, f32(P["ln1_w"]), f32(P["ln1_b"]), spec.eps)
            st2 = (m2_, r2)
        st_f = None
        fold = spec.kind == "m2" and FFN_FOLD and KEEP_FFN_NORM and P["b1"] is not None and P["b2"] is not None and d <= 2048
        if fold:
            # fc1's epilogue: z = gelu(u), gelu'(u) and the row statistics of z; fc2 consumes z against W2 diag(gamma) and applies (mu, rstd) in ITS epilogue
            g_n, u, stats = ops.ffn_fc1_fwd(h2, compute_copy(P["w1"]), f32(P["b1"]), spec.act, spec.eps)   # (u holds gelu'(u) on this path)
            st_f = (stats, None)
        elif spec.kind == "m2":
            # fc1 -> [gelu -> ffn_layernorm] fused: gelu(u) never goes to HBM
            u = ops.gemm(h2, compute_copy(P["w1"]), bias=f32(P["b1"]))
            g_n, mf, rf = ops.layernorm_fwd(u, f32(P["ffn_w"]), f32(P["ffn_b"]), spec.eps, act=spec.act)
            st_f = (mf, rf)
        else:
            # u: the pre-activation -- or, when the activation output itself is kept for backward (KEEP_FFN_NORM), act'(pre-activation): the
            # backward then needs no transcendental in the dgrad epilogue (the GELU-derivative epilogue ran at half the speed of the plain ones)
            u = torch.empty(T, P["w1"].shape[0], dtype=BF, device=dev)
            g_n = ops.gemm(h2, compute_copy(P["w1"]), bias=f32(P["b1"]), act=spec.act, aux=u, aux_grad=KEEP_FFN_NORM)
        res = mid if pre_ln else h2
        if fold:
            w2g, _, c_w2g, b2f = _folded_w2(P)
            y = ops.ffn_fc2_fwd(g_n, w2g, c_w2g, b2f, stats, res)
        elif p_hid > 0:
            y = ops.dropout_add(ops.gemm(g_n, compute_copy(P["w2"]), bias=f32(P["b2"])), p_hid, seed + 2, residual=res)
        else:
            y = ops.gemm(g_n, compute_copy(P["w2"]), bias=f32(P["b2"]), residual=res)
        kept_gn = g_n if KEEP_FFN_NORM else None   # m2: ffn_layernorm(gelu(u)); clip / bert: act(u) -- either way fc2's wgrad operand
        del g_n
        st_y = None
        s2 = None
        if not pre_ln:
            s2 = y
            y, my, ry = ops.layernorm_fwd(s2, f32(P["ln2_w"]), f32(P["ln2_b"]), spec.eps)
            st_y = (my, ry)
        saved = [x2, qkv, o, lse, mid, u, key_bias]
        for st in (st1, st_in, st2, st_f, st_y):
            saved += list(st) if st is not None else [None, None]
        saved.append(s2)
        saved.append(kept_gn)
        y3 = y.view(B, N, d)
        saved.append(y3 if fold else None)   # the fold's backward takes one of the LayerNorm's row means from the layer output (an output may be saved)
        saved.append(h2 if (KEEP_LN2_OUT and pre_ln) else None)
        ctx.save_for_backward(*saved, *params)
        ctx.spec, ctx.shape, ctx.nsaved, ctx.drop = spec, (B, N, d), len(saved), (p_att, p_hid, seed)
        ctx.u_is_grad = bool(KEEP_FFN_NORM and spec.kind != "m2")
        ctx.fold = fold
        return y3

    @staticmethod
    def backward(ctx, dy):
        spec = ctx.spec
        B, N, d = ctx.shape
        T = B * N
        sv = ctx.saved_tensors
        x2, qkv, o, lse, mid, u, key_bias = sv[:7]
        (m1, r1, mi, ri, m2_, r2, mf, rf, my, ry) = sv[7:17]
        s2, kept_gn, y_out, kept_h2 = sv[17], sv[18], sv[19], sv[20]
        params = sv[ctx.nsaved:]
        P = dict(zip(SLOTS, params))
        sink = GradSink()
        pre_ln = spec.kind in ("clip", "m2")
        scale = 64 ** -0.5
        p_att, p_hid, seed = ctx.drop
        dy2 = dy.reshape(T, d)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()

        def lnw(slot):  # accumulate LayerNorm parameter grads
            w, b = P[slot + "_w"], P[slot + "_b"]
            return (sink.buf(w) if w.requires_grad else None), (sink.buf(b) if b.requires_grad else None)

        # ---- MLP half
        b2_fused = False
        if pre_ln:
            ds2 = dy2
        else:
            dgw, dgb = lnw("ln2")
            # (post-LN layers without hidden dropout: the fc2 output's gradient IS this LayerNorm backward's dx -- its column sums, fc2's bias gradient, come out of the same kernel)
            b2_fused = p_hid == 0 and not ctx.fold and P["b2"] is not None and P["b2"].requires_grad
            ds2 = ops.layernorm_bwd(dy2, s2, my, ry, f32(P["ln2_w"]), dgw, dgb, dxsum=sink.buf(P["b2"]) if b2_fused else None)
        if ctx.fold:
            g_n = None
        elif kept_gn is not None:
            g_n = kept_gn
        elif spec.kind == "m2":
            g_n, _, _ = ops.layernorm_fwd(u, f32(P["ffn_w"]), f32(P["ffn_b"]), spec.eps, want_stats=False, act=spec.act)
        else:
            g_n = ops.act_fwd(u, spec.act)
        # hidden dropout: the dense output's gradient is the masked / rescaled ds2; the residual branch keeps ds2 itself
        dy_w2 = ops.dropout_add(ds2.contiguous(), p_hid, seed + 2) if p_hid > 0 else ds2
        if not ctx.fold:
            _wgrad(sink, P["w2"], dy_w2, g_n)
        # fc2's bias gradient = column sums of the incoming gradient.  When that gradient is the dx of the NEXT layer's ln1 backward, that
        # kernel has already summed its columns (handed over on the tensor, valid only while the tensor is unmodified: _version check)
        handed = getattr(dy, "_antmmf_colsum", None) if (pre_ln and p_hid == 0 and not DEBUG_NO_HANDOFF) else None
        handed_ok = handed is not None and handed[1:] == (dy._version, dy.data_ptr(), tuple(dy.shape)) and handed[0].shape[0] == d
        if ctx.fold:
            # ---- sub-LN fold: one row pass over d-wide tensors (the LayerNorm's two row means, rstd-scaled dy for fc2's wgrad, the column sums the
            # LayerNorm parameters need), then everything 4d-wide happens in GEMM epilogues
            z, dact, stats = kept_gn, u, mf
            w2g, w2gt, c_w2g, b2f = _folded_w2(P)
            s_col = torch.zeros(d, dtype=torch.float32, device=dy2.device)
            cs_col = None if handed_ok else torch.zeros(d, dtype=torch.float32, device=dy2.device)
            rowv4, dys = ops.ffn_bwd_rows(ds2, y_out.view(T, d), mid, b2f, c_w2g, stats, z.shape[1], s_col, cs_col)
            cs = handed[0] if handed_ok else cs_col
            if handed_ok:
                COLSUM_HANDOFFS[0] += 1
            if P["b2"].requires_grad:
                sink.buf(P["b2"]).add_(cs)      # b2f = b2 + W2 beta: d b2 = column sums of dy
            b1_fused = True
            du = ops.ffn_fc2_dgrad(ds2, w2gt, z, dact, rowv4, sink.buf(P["b1"]) if P["b1"].requires_grad else None)
            if P["w2"].requires_grad or P["ffn_w"].requires_grad or P["ffn_b"].requires_grad:
                Gm = torch.zeros(d, z.shape[1], dtype=torch.float32, device=dy2.device)
                ops.gemm_wgrad_(Gm, dys, z)
                dW2 = sink.buf(P["w2"]) if P["w2"].requires_grad else torch.empty_like(Gm)
                ops.ffn_wgrad_post_(dW2, Gm, f32(P["w2"]), f32(P["ffn_w"]), f32(P["ffn_b"]), s_col, cs,
                                    sink.buf(P["ffn_w"]) if P["ffn_w"].requires_grad else None, sink.buf(P["ffn_b"]) if P["ffn_b"].requires_grad else None)
                del Gm
            del dys, z, dact
        elif (handed_ok and P["b2"] is not None and P["b2"].requires_grad):
            sink.buf(P["b2"]).add_(handed[0])
            COLSUM_HANDOFFS[0] += 1
        elif not b2_fused:
            _bgrad(sink, P["b2"], dy_w2)
        del g_n
        if ctx.fold:
            pass
        elif spec.kind == "m2":
            dgn = dgrad(ds2, P["w2"])
            dgw, dgb = lnw("ffn")
            # through LN and gelu at once; the fc1 bias gradient (column sums of du) falls out of the same pass
            b1_fused = P["b1"] is not None and P["b1"].requires_grad
            du = ops.layernorm_bwd(dgn, u, mf, rf, f32(P["ffn_w"]), dgw, dgb, act=spec.act, dxsum=sink.buf(P["b1"]) if b1_fused else None)
            del dgn
        else:
            b1_fused = False
            w2t = compute_copy_t(P["w2"]) if isinstance(P["w2"], torch.nn.Parameter) else None
            if (GATED_DGRAD_COLSUM and ctx.u_is_grad and w2t is not None and P["b1"] is not None and P["b1"].requires_grad and dy_w2.is_contiguous()
                    and ops.gemm_gated_colsum_ok(T, w2t.shape[0], w2t.shape[1])):
                # (d(dense out) W2) * act'(u) with the column sums of the result as 256 partial rows out of the same kernel: fc1's bias gradient = their column sums --
                # no pass over the 4d-wide du (0.39 ms per cross-encoder layer of the video workloads)
                du, parts = ops.gemm_gated_colsum(dy_w2, w2t, u, act=spec.act)
                ops.colsum_(sink.buf(P["b1"]), parts)
                b1_fused = True
            else:
                du = dgrad(dy_w2, P["w2"], gate=u, act=spec.act, gate_is_grad=ctx.u_is_grad)  # (d(dense out) W2) * act'(u)
        # The normalised tensors the wgrads need (h2, o_n, h) are not kept by the forward pass and not recomputed either: the LayerNorm
        # backward of the same tensor re-emits them (layernorm_bwd_renorm: one extra write instead of a read + write pass).
        ln_mid = ("ln2" if pre_ln else "ln1")
        dgw, dgb = lnw(ln_mid)
        bo_fused = P["bo"] is not None and P["bo"].requires_grad and p_hid == 0  # out-projection bias gradient = column sums of dmid
        bo_sum = sink.buf(P["bo"]) if bo_fused else None
        if pre_ln:
            dh2 = dgrad(du, P["w1"])
            if kept_h2 is not None:   # (experiment: the normalised tensor was kept -- the plain backward, four streams)
                dmid, h2 = ops.layernorm_bwd(dh2, mid, m2_, r2, f32(P["ln2_w"]), dgw, dgb, dres=ds2, dxsum=bo_sum), kept_h2
            else:
                dmid, h2 = ops.layernorm_bwd_renorm(dh2, mid, m2_, r2, f32(P["ln2_w"]), f32(P["ln2_b"]), dgw, dgb, dres=ds2, dxsum=bo_sum)  # + residual path
            del dh2
        else:
            da = dgrad(du, P["w1"], residual=ds2)  # bert: a feeds the MLP and the residual
            dmid, h2 = ops.layernorm_bwd_renorm(da, mid, m2_, r2, f32(P["ln1_w"]), f32(P["ln1_b"]), dgw, dgb, dxsum=bo_sum)
            del da
        _wgrad(sink, P["w1"], du, h2)
        if not b1_fused:
            _bgrad(sink, P["b1"], du)
        del h2
        del du

        # ---- attention half
        o2 = o.view(T, d)
        dy_wo = ops.dropout_add(dmid.contiguous(), p_hid, seed + 1) if p_hid > 0 else dmid
        do = dgrad(dy_wo, P["wo"])
        bv_fused = False
        if spec.kind == "m2":
            dgw, dgb = lnw("inner")
            # The value-projection bias gradient for free: softmax rows sum to one, so sum_k dV[k] = sum_q dO[q] per head -- the column sums of the gradient at
            # the attention output, which the inner LayerNorm's backward can emit while it writes that tensor (no column-sum pass over dV).  Needs the separate
            # v bias of this layer kind and no attention-probability dropout (dropped rows no longer sum to one).
            bv_fused = (not spec.packed_qkv) and p_att == 0 and P["bv"] is not None and P["bv"].requires_grad
            do, o_n = ops.layernorm_bwd_renorm(do, o2, mi, ri, f32(P["inner_w"]), f32(P["inner_b"]), dgw, dgb, dxsum=sink.buf(P["bv"]) if bv_fused else None)
        else:
            o_n = o2
        _wgrad(sink, P["wo"], dy_wo, o_n)
        if not bo_fused:
            _bgrad(sink, P["bo"], dy_wo)
        del o_n
        q3 = qkv.view(B, N, 3 * d)
        dqkv = torch.empty(B, N, 3 * d, dtype=BF, device=qkv.device)
        # q / k / v bias gradients: the one-kernel attention backward also returns, per batch item, the token sums of dQ | dK | dV ([B, 3d] fp32) -- their column
        # sums are the bias gradients, so the column-sum passes over the B * N rows of dQ | dK | dV (0.19 ms per image-tower layer of the flagship) are not run
        qkv_biases = [P["bqkv"]] if spec.packed_qkv else [P["bq"], P["bk"], P["bv"]]
        tok_sums = None
        # (ATTN_BWD_SUMS_MIN_TOKENS = 129 restricts it to the long towers: the A/B of profiles/r6b_attn_bwd_token_sums_ab.txt -- the short towers gain too, a little)
        if (ATTN_BWD_SUMS and N >= ATTN_BWD_SUMS_MIN_TOKENS and d == 64 * spec.heads and any(b_ is not None and b_.requires_grad for b_ in qkv_biases)
                and ops.attention_bwd_sums_ok(64, N, N, p_att)):
            tok_sums = torch.empty(B, 3 * d, dtype=torch.float32, device=qkv.device)
        ops.attention_bwd(q3[..., :d], q3[..., d:2 * d], q3[..., 2 * d:], o, lse, do.view(B, N, d), spec.heads, scale, key_bias,
                          dq=dqkv[..., :d], dk=dqkv[..., d:2 * d], dv=dqkv[..., 2 * d:], dropout_p=p_att, dropout_seed=seed, sums=tok_sums, sums_v=not bv_fused)
        del do
        dqkv2 = dqkv.view(T, 3 * d)
        bsrc = tok_sums if tok_sums is not None else dqkv2   # what the q / k / v bias gradients are column sums of
        dx = None
        if pre_ln:   # (also when x itself needs no gradient: ln1's own parameters do)
            dh = ops.gemm(dqkv2, _packed_qkv_weight_t(P, spec))
            dgw, dgb = lnw("ln1")
            dx_colsum = torch.zeros(d, dtype=torch.float32, device=dh.device) if ctx.needs_input_grad[0] else None
            dx, h = ops.layernorm_bwd_renorm(dh, x2, m1, r1, f32(P["ln1_w"]), f32(P["ln1_b"]), dgw, dgb, dres=dmid, dxsum=dx_colsum)
            del dh
        else:
            h = x2
        if spec.packed_qkv:
            _wgrad(sink, P["wqkv"], dqkv2, h)
            _bgrad(sink, P["bqkv"], bsrc)
        else:
            ws_ = [P["w" + nm] for nm in "qkv"]
            seg_tiles = 3 * d * d // 65536   # 256 x 256 tiles of the merged launch; it runs tiles x (256 // tiles) workgroups
            if (all(w is not None and w.requires_grad for w in ws_) and d % 256 == 0 and T >= 4096 and T % 64 == 0
                    and (T <= 131072 or seg_tiles * (256 // max(1, seg_tiles)) >= QKV_WGRAD_MERGE_MIN_WGS)):
                # separate q / k / v projections (BERT, torchscale): ONE wgrad GEMM over the packed dQ | dK | dV, its reduce launch scatters the three row segments into the
                # three parameters' gradient buffers.  Measured in the l14 step (profiles/r6_gemm_table_l14_qkv_wgrad_merged.txt): 3072 x 1024 is 48 tiles x 5 token splits
                # = 240 workgroups against 3 x (16 tiles x 16 splits = 256) -- on the 77-token text tower (78848 tokens: 77 K-tiles per split before) 402 us instead of
                # 3 x 150 us, on the image tower (263168 tokens) 1359 us instead of 3 x 445 us: the merged launch leaves 16 CUs idle and loses; hence the token bound
                ops.gemm_wgrad_seg_([sink.buf(w) for w in ws_], dqkv2, h)
            elif (QK_WGRAD_MERGE and all(w is not None and w.requires_grad for w in ws_) and d % 256 == 0 and T >= 4096 and T % 64 == 0
                    and (2 * d * d // 65536) * (256 // max(1, 2 * d * d // 65536)) >= QKV_WGRAD_MERGE_MIN_WGS):
                # (d = 1024 on the long tower: all three merged would run 240 workgroups, q | k merged runs 32 tiles x 8 splits = all 256 with twice the K depth per workgroup; v on its own)
                ops.gemm_wgrad_seg_([sink.buf(ws_[0]), sink.buf(ws_[1])], dqkv2[:, :2 * d], h)
                _wgrad(sink, P["wv"], dqkv2[:, 2 * d:], h)
            else:
                for i, nm in enumerate("qkv"):
                    _wgrad(sink, P["w" + nm], dqkv2[:, i * d:(i + 1) * d], h)
            rqk = _arena_run([P["bq"], P["bk"]]) if (bv_fused and P["bq"] is not None and P["bk"] is not None and P["bq"].requires_grad and P["bk"].requires_grad) else None
            if rqk is not None:
                # q and k biases adjacent in the gradient arena (arena.tag_pack): ONE column-sum launch over dQ | dK into both slots (the v bias comes out of the inner LayerNorm's backward)
                ops.colsum_(rqk[0].grad[rqk[1]:rqk[1] + 2 * d], bsrc[:, :2 * d])
            else:
                rqkv = _arena_run([P["bq"], P["bk"], P["bv"]]) if (tok_sums is not None and not bv_fused and all(P["b" + nm] is not None and P["b" + nm].requires_grad for nm in "qkv")) else None
                if rqkv is not None:   # (BERT: the three biases adjacent in the gradient arena -- one launch over the [B, 3d] sums)
                    ops.colsum_(rqkv[0].grad[rqkv[1]:rqkv[1] + 3 * d], bsrc)
                else:
                    for i, nm in enumerate("qkv"):
                        if not (nm == "v" and bv_fused):
                            _bgrad(sink, P["b" + nm], bsrc[:, i * d:(i + 1) * d])
        del h
        if ctx.needs_input_grad[0]:
            if not pre_ln:
                dx = ops.gemm(dqkv2, _packed_qkv_weight_t(P, spec), residual=dmid)
            dx = dx.view(B, N, d)
            if pre_ln:
                # for the previous layer's fc2 bias gradient (see above): valid only for this very tensor, unmodified
                dx._antmmf_colsum = (dx_colsum, dx._version, dx.data_ptr(), tuple(dx.shape))
        else:
            dx = None
        grads = [sink.result(p, p is not None and ctx.needs_input_grad[4 + i]) for i, p in enumerate(params)]
        arena = next((getattr(p, "_antmmf_arena", None) for p in params if p is not None), None)
        if arena is not None:
            arena.note_backward(params)   # these parameters' gradients may now be final: their bucket's all-reduce can start
        return (dx, None, None, None, *grads)


def transformer_layer(x, spec, params, key_bias=None, seed=None):
    """x [B, N, d] bf16 -> [B, N, d]; `params` maps SLOTS names to fp32 master parameters.  `seed`: dropout seed of this call
    (BERT layers with dropout > 0 only); None draws one from torch's CPU generator (host side: no device sync)."""
    if seed is None and spec.kind == "bert" and (spec.attn_dropout > 0 or spec.hidden_dropout > 0):
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    plist = [params.get(s) for s in SLOTS]
    arena = next((getattr(p, "_antmmf_arena", None) for p in plist if p is not None), None)
    if arena is not None and torch.is_grad_enabled():
        arena.note_forward(plist)   # gradient all-reduce under the backward pass (arena.arm_overlap): counts this use of the parameters
    return _TransformerLayer.apply(x, key_bias, spec, seed, *plist)


# ------------------------------------------------------------------------------ embeddings
class _PatchEmbed(torch.autograd.Function):
    """image [B, 3, H, W] -> tokens [B, G+1, d]: patch GEMM (conv with stride == kernel), [cls] row, + positional
    embedding (clip/model.py:310-323; torchscale VisionEmbedding embedding.py:67-83 + positions 2.. :92-110)."""

    @staticmethod
    def forward(ctx, image, weight, bias, cls, pos, patch, shift, scale):
        B = image.shape[0]
        d = weight.shape[0]
        kk = weight[0].numel()
        kpad = ((kk + 63) // 64) * 64
        patches = ops.patchify(image.contiguous(), patch, kpad, shift, scale)
        w2 = compute_copy(weight).reshape(d, kk)
        if kpad != kk:
            w2 = torch.nn.functional.pad(w2, (0, kpad - kk))
        tok = ops.gemm(patches, w2.contiguous())
        G = patches.shape[0] // B
        x = ops.assemble_tokens(tok, f32(cls).reshape(-1).contiguous(), f32(pos).contiguous() if pos is not None else None,
                                f32(bias), B, G)
        ctx.save_for_backward(patches, weight, bias, cls, pos)
        ctx.dims = (B, G, d, kk, kpad)
        return x

    @staticmethod
    def backward(ctx, dx):
        patches, weight, bias, cls, pos = ctx.saved_tensors
        B, G, d, kk, kpad = ctx.dims
        dx = dx.contiguous()
        sink = GradSink()
        dtok = ops.split_tokens(dx)
        if weight.requires_grad:
            dwp = torch.zeros(d, kpad, dtype=torch.float32, device=dx.device)
            # (kpad = 640 for 14 x 14 patches is not a multiple of 256: the 128 x 128-tile kernel, 40 tiles -- split over the tokens to cover the chip)
            ops.gemm(dtok, patches, out=dwp, p_rmajor=True, q_rmajor=True, accumulate=True, split_k=min(16, max(4, dtok.shape[0] // 16384)))
            sink.buf(weight).add_(dwp[:, :kk].reshape(weight.shape))
        _bgrad(sink, bias, dtok)
        if cls.requires_grad:
            ops.colsum_(sink.buf(cls).view(-1), dx.view(B, (G + 1) * d)[:, :d])
        if pos is not None and pos.requires_grad:
            ops.colsum_(sink.buf(pos).view(-1), dx.view(B, (G + 1) * d))
        ng = ctx.needs_input_grad
        return (None, sink.result(weight, ng[1]), sink.result(bias, bias is not None and ng[2]), sink.result(cls, ng[3]),
                sink.result(pos, pos is not None and ng[4]), None, None, None)


def patch_embed(image, weight, bias, cls, pos, patch, shift=0.0, scale=1.0):
    _note_untracked(weight, bias, cls, pos)
    return _PatchEmbed.apply(image, weight, bias, cls, pos, patch, shift, scale)


class _Embed(torch.autograd.Function):
    """Sum of word / position / token-type lookups -> bf16 (BertEmbeddings clip_text_encoder.py:36-60 before its
    LayerNorm; torchscale TextEmbedding + PositionalEmbedding with padded rows zeroed, encoder.py:350-386,440)."""

    @staticmethod
    def forward(ctx, ids, word, pos, type_table, zero_rows, pos_offset, padding_idx):
        out = ops.embed_gather(ids.contiguous(), f32(word), f32(pos), f32(type_table), None, zero_rows, pos_offset)
        ctx.save_for_backward(ids, word, pos, type_table, zero_rows)
        ctx.pos_offset, ctx.padding_idx = pos_offset, padding_idx
        return out

    @staticmethod
    def backward(ctx, dx):
        ids, word, pos, type_table, zero_rows = ctx.saved_tensors
        dx = dx.contiguous()
        seq = ids.shape[1]
        sink = GradSink()
        if word.requires_grad:
            # nn.Embedding(padding_idx=k) never updates row k (BertEmbeddings: modeling_bert.py:71-73, clip_text_encoder.py:21-23, padding_idx = 0).  Skipping those
            # rows is also what keeps the sorted scatter balanced: on real captions every [PAD] position carries id 0 -- one run of tens of thousands of rows.
            skip = zero_rows
            if ctx.padding_idx is not None:
                pad = (ids == ctx.padding_idx).to(torch.uint8)
                skip = pad if skip is None else (skip.to(torch.uint8).reshape(pad.shape) | pad)
            ops.embed_scatter_add_(sink.buf(word), dx, ids.contiguous(), skip)
        if pos is not None and pos.requires_grad:
            ops.embed_scatter_add_(sink.buf(pos), dx, None, zero_rows, seq=seq, offset=ctx.pos_offset)
        if type_table is not None and type_table.requires_grad:
            # token-type ids are all zero on this path: row 0 gets the column sum
            tmp = torch.zeros(dx.shape[-1], dtype=torch.float32, device=dx.device)
            d2 = dx.view(-1, dx.shape[-1])
            if zero_rows is not None:
                d2 = d2 * (1 - zero_rows.view(-1, 1).to(d2.dtype))
            ops.colsum_(tmp, d2.contiguous())
            sink.buf(type_table)[0].add_(tmp)
        ng = ctx.needs_input_grad
        return (None, sink.result(word, ng[1]), sink.result(pos, pos is not None and ng[2]),
                sink.result(type_table, type_table is not None and ng[3]), None, None, None)


def embed(ids, word, pos=None, type_table=None, zero_rows=None, pos_offset=0, padding_idx=None):
    """padding_idx: table row that receives no gradient (torch.nn.Embedding's padding_idx)."""
    _note_untracked(word, pos, type_table)
    return _Embed.apply(ids, word, pos, type_table, zero_rows, pos_offset, padding_idx)
