"""ViLBERT co-attention operator on the MI355X path (reference: antmmf/models/vilbert.py:285-416 `BertBiAttention`).

Only the cross-modal co-attention OPERATOR of the `vilbert` registry model is on the contrastive path's scope table (operator-level target);
the rest of ViLBERT (embeddings, the two single-stream stacks, pooling, task heads) is another model family.  Same class name, constructor
fields (`bi_hidden_size`, `bi_num_attention_heads`, `v_hidden_size`, `hidden_size`, `v_attention_probs_dropout_prob`,
`attention_probs_dropout_prob`, `visualization`), parameter names (query1 / key1 / value1 / query2 / key2 / value2: state_dicts map 1:1) and
forward signature / return triple as the reference.

MI355X design: per stream ONE packed [3 A, d_in] projection GEMM (the reference issues three), then the fused attention kernel twice with
the streams swapped -- stream-2 queries over stream-1 keys / values under stream 1's additive mask, and vice versa -- with the
attention-probability dropout of each direction applied INSIDE the kernel from a counter-based mask (nothing [B, h, Nq, Nk]-shaped reaches HBM).
The additive masks are the reference's extended masks `[B, 1, 1, N]` (0 / -10000); they are passed to the kernel as per-key biases [B, N].
Head size: 64 (every tower of the contrastive path) or 128 (ViLBERT's own bi_hidden_size 1024 / 8 heads); another
`bi_hidden_size / bi_num_attention_heads` raises NotImplementedError.  `visualization=True` (returning the probability tensors) is not available from the fused kernel and raises."""
import math

import torch
from torch import nn

from antmmf.hip import functional as HF
from antmmf.hip import ops


class _AttentionDrop(torch.autograd.Function):
    """Fused attention with attention-probability dropout (counter-based mask regenerated in backward)."""

    @staticmethod
    def forward(ctx, q, k, v, key_bias, heads, scale, p, seed):
        o, lse = ops.attention_fwd(q, k, v, heads, scale, key_bias, dropout_p=p, dropout_seed=seed)
        ctx.save_for_backward(q, k, v, o, lse, key_bias)
        ctx.meta = (heads, scale, p, seed)
        return o

    @staticmethod
    def backward(ctx, d_o):
        q, k, v, o, lse, key_bias = ctx.saved_tensors
        heads, scale, p, seed = ctx.meta
        dq, dk, dv = ops.attention_bwd(q, k, v, o, lse, d_o.contiguous(), heads, scale, key_bias, dropout_p=p, dropout_seed=seed)
        return dq, dk, dv, None, None, None, None, None


class BertBiAttention(nn.Module):
    _antmmf_hip_native = True

    def __init__(self, config):
        super().__init__()
        if config.bi_hidden_size % config.bi_num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                             % (config.bi_hidden_size, config.bi_num_attention_heads))
        self.visualization = bool(config.get("visualization", False))
        self.num_attention_heads = config.bi_num_attention_heads
        self.attention_head_size = int(config.bi_hidden_size / config.bi_num_attention_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        if self.attention_head_size not in (64, 128):
            raise NotImplementedError(f"BertBiAttention on the HIP path: head size {self.attention_head_size} (the attention kernels are built for 64 and 128)")
        self.query1 = nn.Linear(config.v_hidden_size, self.all_head_size)
        self.key1 = nn.Linear(config.v_hidden_size, self.all_head_size)
        self.value1 = nn.Linear(config.v_hidden_size, self.all_head_size)
        self.dropout1 = nn.Dropout(config.v_attention_probs_dropout_prob)
        self.query2 = nn.Linear(config.hidden_size, self.all_head_size)
        self.key2 = nn.Linear(config.hidden_size, self.all_head_size)
        self.value2 = nn.Linear(config.hidden_size, self.all_head_size)
        self.dropout2 = nn.Dropout(config.attention_probs_dropout_prob)

    def _qkv(self, x, q, k, v):
        """packed projection of one stream: [B, N, d_in] -> three [B, N, A] views of one [B, N, 3 A] buffer."""
        w = torch.cat([q.weight, k.weight, v.weight], 0)
        b = torch.cat([q.bias, k.bias, v.bias], 0)
        y = HF.linear(x.to(torch.bfloat16) if x.dtype != torch.bfloat16 else x, w, b)
        a = self.all_head_size
        return y[..., :a], y[..., a:2 * a], y[..., 2 * a:]

    @staticmethod
    def _key_bias(mask, n):
        """extended additive mask [B, 1, 1, N] (or [B, N]) -> fp32 [B, N]"""
        if mask is None:
            return None
        m = mask.reshape(mask.shape[0], -1).float()
        assert m.shape[1] == n, "co-attention takes per-key additive masks ([B, 1, 1, N]); per-query masks are not supported"
        return m.contiguous()

    def forward(self, input_tensor1, attention_mask1, input_tensor2, attention_mask2, co_attention_mask=None, use_co_attention_mask=False,
                dropout_seeds=None):
        if self.visualization:
            raise NotImplementedError("visualization=True needs the probability tensors, which the fused kernel never materialises")
        q1, k1, v1 = self._qkv(input_tensor1, self.query1, self.key1, self.value1)   # vision stream
        q2, k2, v2 = self._qkv(input_tensor2, self.query2, self.key2, self.value2)   # text stream
        scale = 1.0 / math.sqrt(self.attention_head_size)
        b1 = self._key_bias(attention_mask1, input_tensor1.shape[1])
        b2 = self._key_bias(attention_mask2, input_tensor2.shape[1])
        p1 = self.dropout1.p if self.training else 0.0
        p2 = self.dropout2.p if self.training else 0.0
        if dropout_seeds is None and (p1 > 0 or p2 > 0):
            dropout_seeds = [int(s) for s in torch.randint(0, 2 ** 62, (2,)).tolist()]   # host generator: no device sync
        s1, s2 = dropout_seeds if dropout_seeds is not None else (0, 0)
        h = self.num_attention_heads
        # context 1: text queries over vision keys / values (reference :357-376); context 2: vision queries over text (:378-401)
        ctx1 = _AttentionDrop.apply(q2, k1, v1, b1, h, scale, p1, s1) if p1 > 0 else HF.attention(q2, k1, v1, h, scale, b1)
        ctx2 = _AttentionDrop.apply(q1, k2, v2, b2, h, scale, p2, s2) if p2 > 0 else HF.attention(q1, k2, v2, h, scale, b2)
        return ctx1, ctx2, None
