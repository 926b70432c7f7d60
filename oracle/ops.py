"""oracle.ops -- CPU restatement (plain torch fp32) of the elementwise / attention operators on the
contrastive hot path.

TEST INFRASTRUCTURE ONLY.  This package is the checker for the HIP path: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import it.  Nothing under
ant-multi-modal-framework_amd/ may import it.  It is pinned against fixtures under tests/golden/
that were produced by executing the reference implementation (tests/golden/make_golden.py).

Every function cites the reference file:line whose arithmetic it restates (paths relative to the
reference checkout).
"""
import math

import torch
import torch.nn.functional as F


def layer_norm(x, weight, bias, eps):
    """Row LayerNorm computed in fp32 and cast back (CLIP: antmmf/modules/vision/backbone/clip/model.py:213-219;
    BERT: modeling_bert.py:63 with eps 1e-12; M2: torchscale LayerNorm eps 1e-5, encoder.py:34)."""
    xf = x.float()
    mu = xf.mean(-1, keepdim=True)
    var = ((xf - mu) ** 2).mean(-1, keepdim=True)
    y = (xf - mu) * torch.rsqrt(var + eps)
    if weight is not None:
        y = y * weight.float() + bias.float()
    return y.to(x.dtype)


def quick_gelu(x):
    """x * sigmoid(1.702 x)  (clip/model.py:222-224)."""
    return x * torch.sigmoid(1.702 * x)


def gelu_erf(x):
    """0.5 x (1 + erf(x / sqrt 2))  (modeling_bert.py:31-37; torchscale F.gelu, feedforward_network.py:80-86)."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def linear(x, w, b=None):
    y = x @ w.t()
    return y if b is None else y + b


def split_heads(t, heads):
    """[B, N, h*dh] -> [B, h, N, dh]"""
    b, n, d = t.shape
    return t.view(b, n, heads, d // heads).permute(0, 2, 1, 3)


def merge_heads(t):
    """[B, h, N, dh] -> [B, N, h*dh]"""
    b, h, n, dh = t.shape
    return t.permute(0, 2, 1, 3).reshape(b, n, h * dh)


def attention_core(q, k, v, scale, key_bias=None):
    """softmax(scale * q k^T + key_bias) v per head, softmax in fp32.

    q,k,v: [B, h, N, dh].  key_bias: [B, Nk] additive fp32 term per key (BERT uses -10000 on padded
    keys, modeling_bert.py:144-161 & clip_text_encoder.py:97-100; torchscale uses -inf,
    multihead_attention.py:130-142; CLIP's ViT uses none, clip/model.py:245-251)."""
    s = torch.matmul(q, k.transpose(-1, -2)).float() * scale
    if key_bias is not None:
        s = s + key_bias[:, None, None, :].float()
    p = torch.softmax(s, dim=-1).to(q.dtype)
    return torch.matmul(p, v)


def clip_mha(x, in_w, in_b, out_w, out_b, heads, key_bias=None):
    """nn.MultiheadAttention self-attention as CLIP uses it, on [B, N, d] (the reference feeds LND;
    the arithmetic per (batch, head) is identical).  Packed in_proj [3d, d] = (q | k | v) rows;
    q is scaled by dh^-0.5.  clip/model.py:231,245-251."""
    b, n, d = x.shape
    qkv = linear(x, in_w, in_b)
    q, k, v = qkv.split(d, dim=-1)
    dh = d // heads
    ctx = attention_core(split_heads(q, heads), split_heads(k, heads), split_heads(v, heads), dh ** -0.5, key_bias)
    return linear(merge_heads(ctx), out_w, out_b)


def l2_normalize(x, eps=1e-12):
    """F.normalize(x, p=2, dim=-1): x / max(||x||, eps)  (univl_video_base.py:114,158)."""
    return x / x.norm(dim=-1, keepdim=True).clamp_min(eps)


def l2_normalize_noeps(x):
    """x / ||x||  (M2: vlmo_module.py:346,349,394,397 divide by the raw norm)."""
    return x / x.norm(dim=-1, keepdim=True)


def co_attention(q1, k1, v1, q2, k2, v2, bias1=None, bias2=None):
    """ViLBERT co-attention core (antmmf/models/vilbert.py:360-400): stream-1 context is computed
    from stream-2 queries against stream-1 keys/values and vice versa.
    q*, k*, v*: [B, h, N*, dh]; bias*: [B, N*] additive per key of that stream.
    Returns (ctx_for_stream2_queries_over_stream1, ctx_for_stream1_queries_over_stream2)."""
    dh = q1.shape[-1]
    ctx1 = attention_core(q2, k1, v1, 1.0 / math.sqrt(dh), bias1)
    ctx2 = attention_core(q1, k2, v2, 1.0 / math.sqrt(dh), bias2)
    return ctx1, ctx2
