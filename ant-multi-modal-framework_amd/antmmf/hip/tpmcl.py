"""Autograd wrappers of the DMAE stage-3 head kernels (csrc/tpmcl.hip) and an fp32-accurate Linear on the MFMA pipe.

What they replace in the reference (prj/dmae_vtp/roi_univl/univl/model/dmae_utils.py, tpmcl_utils.py):
  token_weights     Linear(D, 1) + masked_fill(-inf) + softmax over the tokens (text_weight_fc / video_weight_fc, :147-165)
  pair_dots         einsum('ctd,cvd->ctv') on aligned pairs with one text token (wti_interaction_row, :425-470)
  pair_wsum         einsum('abd,ab->ad') (the predicted global text feature, :411-418)
  tis_keep          sort / cumsum / scatter of TokenImportanceSelector (tpmcl_utils.py:101-121); not differentiable (a mask)
  linear_f32        the small fp32 matmuls of the weight predictors (tpmcl_utils.py:35-50), on the bf16 MFMA GEMM via the hi / lo split of
                    contrastive.matmul_f32 (three GEMMs: fp32-accurate products, fp32 accumulation)
No torch / rocBLAS GEMM, softmax or sort is left on that path.
"""
import torch

from . import ops
from .contrastive import matmul_f32      # fp32-accurate products on the bf16 MFMA GEMM (hi / lo split); module-level so that the lane-emulator tests can stub it
from .functional import GradSink, _note_untracked, f32

TOKEN_WEIGHT_MAX_T, TOKEN_WEIGHT_MAX_D, TIS_KEEP_MAX_T = 128, 1024, 64   # limits of csrc/tpmcl.hip's row kernels


class _TokenWeights(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, weight, bias, mask):
        feat = feat.float().contiguous()
        m = None if mask is None else mask.float().contiguous()
        w = f32(weight).reshape(-1).contiguous()
        p = ops.token_weight_fwd(feat, w, None if bias is None else f32(bias).reshape(-1), m)
        ctx.save_for_backward(feat, weight, bias, p)
        return p

    @staticmethod
    def backward(ctx, dp):
        feat, weight, bias, p = ctx.saved_tensors
        sink = GradSink()
        dw = sink.buf(weight) if weight.requires_grad else torch.zeros_like(weight, dtype=torch.float32)
        db = (sink.buf(bias) if bias.requires_grad else None) if bias is not None else None
        dfeat = ops.token_weight_bwd(feat, f32(weight).reshape(-1).contiguous(), p, dp.float().contiguous(), dw.view(-1),
                                     None if db is None else db.view(-1), want_dfeat=ctx.needs_input_grad[0])
        return (dfeat, sink.result(weight, ctx.needs_input_grad[1]), sink.result(bias, bias is not None and ctx.needs_input_grad[2]), None)


def token_weights(feat, weight, bias=None, mask=None):
    """feat [N, T, D] -> softmax over T of (feat . weight + bias), tokens with mask < 0.5 excluded.  weight: the [1, D] (or [D]) parameter of
    an nn.Linear(D, 1), bias its [1] bias."""
    _note_untracked(weight, bias)
    if feat.shape[-2] > TOKEN_WEIGHT_MAX_T or feat.shape[-1] > TOKEN_WEIGHT_MAX_D:
        # beyond the fused kernel's row budget (one workgroup holds a row's T x D features: csrc/tpmcl.hip) -- e.g. frames x patches video tokens: the same
        # arithmetic as device tensor ops (a shape guard for configurations the shipped ymls do not use, not a second product path)
        logits = (feat.float() * weight.float().reshape(-1)).sum(-1)
        if bias is not None:
            logits = logits + bias.float().reshape(())
        if mask is not None:
            logits = logits.masked_fill(mask < 0.5, float("-inf"))
        return torch.softmax(logits, dim=-1)
    return _TokenWeights.apply(feat, weight, bias, mask)


class _PairDots(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        x, y = x.float().contiguous(), y.float().contiguous()
        ctx.save_for_backward(x, y)
        return ops.pair_dots(x, y)

    @staticmethod
    def backward(ctx, dout):
        x, y = ctx.saved_tensors
        dout = dout.float().contiguous()
        dx = ops.pair_wsum(dout, y) if ctx.needs_input_grad[0] else None
        dy = ops.pair_outer(dout, x) if ctx.needs_input_grad[1] else None
        return dx, dy


def pair_dots(x, y):
    """x [C, D], y [C, V, D] -> [C, V]: x[c] . y[c, v]."""
    if y.shape[-2] > TOKEN_WEIGHT_MAX_T:   # its backward (pair_wsum / pair_outer) holds the V weights of a pair in one workgroup: same shape guard as token_weights
        return torch.einsum("cd,cvd->cv", x.float(), y.float())
    return _PairDots.apply(x, y)


class _PairWsum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, y):
        w, y = w.float().contiguous(), y.float().contiguous()
        ctx.save_for_backward(w, y)
        return ops.pair_wsum(w, y)

    @staticmethod
    def backward(ctx, dout):
        w, y = ctx.saved_tensors
        dout = dout.float().contiguous()
        dw = ops.pair_dots(dout, y) if ctx.needs_input_grad[0] else None
        dy = ops.pair_outer(w, dout) if ctx.needs_input_grad[1] else None
        return dw, dy


def pair_wsum(w, y):
    """w [C, V], y [C, V, D] -> [C, D]: sum_v w[c, v] y[c, v]."""
    if y.shape[-2] > TOKEN_WEIGHT_MAX_T:
        return torch.einsum("cv,cvd->cd", w.float(), y.float())
    return _PairWsum.apply(w, y)


def tis_keep(weights, thresh):
    """[R, T] token weights -> keep mask (0 = one of the most important tokens whose descending cumulative weight is < thresh)."""
    w = weights.detach().float().contiguous()
    if w.shape[-1] > TIS_KEEP_MAX_T:   # rows longer than the kernel's 64 lanes: the reference's sort / cumsum / scatter (tpmcl_utils.py:101-121) on the device
        order = torch.argsort(w, dim=-1, descending=True)
        drop_sorted = torch.cumsum(torch.gather(w, -1, order), dim=-1) < thresh
        return 1.0 - torch.zeros_like(w).scatter_(-1, order, drop_sorted.float())
    return ops.tis_keep(w, float(thresh))


class _LinearF32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).float()
        W = weight.float()
        ctx.save_for_backward(x2, W)
        ctx.shp = shp
        return matmul_f32(x2, W).reshape(*shp[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, W = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1]).float()
        dx = matmul_f32(dy2, W, b_rmajor=True).reshape(ctx.shp) if ctx.needs_input_grad[0] else None   # dx[i, k] = sum_j dy[i, j] W[j, k]
        dW = matmul_f32(dy2, x2, a_rmajor=True, b_rmajor=True) if ctx.needs_input_grad[1] else None      # dW[j, k] = sum_i dy[i, j] x[i, k]
        return dx, dW


def linear_f32(x, weight, bias=None):
    """x [..., K] @ weight[J, K]^T (+ bias) to fp32 accuracy on the bf16 MFMA GEMM (hi / lo operand split)."""
    y = _LinearF32.apply(x, weight)
    return y if bias is None else y + bias.float()
