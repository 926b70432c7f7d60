"""CLIP vision transformer on the MI355X kernels.

Same classes, constructor arguments and parameter names as the reference's
antmmf/modules/vision/backbone/clip/model.py:213-335 (LayerNorm, QuickGELU, ResidualAttentionBlock, Transformer,
VisionTransformer) so reference state_dicts load unchanged; the arithmetic runs through antmmf.hip:
the patch conv is a patchify + MFMA GEMM, each ResidualAttentionBlock is ONE fused autograd node
(antmmf.hip.functional.transformer_layer, kind "clip"), activations are bf16 in [B, N, d] layout (the reference
permutes to LND for nn.MultiheadAttention; per-(batch, head) arithmetic is identical).
"""
from collections import OrderedDict

import torch
from torch import nn

from antmmf.hip import functional as HF


class LayerNorm(nn.LayerNorm):
    """fp32-statistics LayerNorm (reference: model.py:213-219)."""

    def forward(self, x):
        return HF.layer_norm(x, self.weight, self.bias, self.eps)


class QuickGELU(nn.Module):
    """x * sigmoid(1.702 x) (reference: model.py:222-224); inside blocks it is fused into the fc GEMM epilogue."""

    def forward(self, x):
        from antmmf.hip import ops

        return ops.act_fwd(x.contiguous(), "quick_gelu") if not x.requires_grad else _quick_gelu_autograd(x)


class _QuickGeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        from antmmf.hip import ops

        x = x.contiguous()
        ctx.save_for_backward(x)
        return ops.act_fwd(x, "quick_gelu")

    @staticmethod
    def backward(ctx, dy):
        from antmmf.hip import ops

        (x,) = ctx.saved_tensors
        return ops.act_bwd(dy.contiguous(), x, "quick_gelu")


def _quick_gelu_autograd(x):
    return _QuickGeluFn.apply(x)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model: int, n_head: int, attn_mask: torch.Tensor = None):
        super().__init__()
        if attn_mask is not None:
            raise NotImplementedError("the vision tower uses no attention mask (reference: model.py:300 passes none)")
        if d_model // n_head != 64:
            raise ValueError("the fused attention kernel is specialised for head_dim 64 (every config on this path)")
        # nn.MultiheadAttention / nn.Linear are used as parameter holders only (same names + init as the reference)
        self.attn = nn.MultiheadAttention(d_model, n_head)
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([
            ("c_fc", nn.Linear(d_model, d_model * 4)),
            ("gelu", QuickGELU()),
            ("c_proj", nn.Linear(d_model * 4, d_model)),
        ]))
        self.ln_2 = LayerNorm(d_model)
        self.attn_mask = None
        self._spec = HF.LayerSpec(kind="clip", heads=n_head, eps=self.ln_1.eps, act="quick_gelu", packed_qkv=True)

    def _params(self):
        return dict(ln1_w=self.ln_1.weight, ln1_b=self.ln_1.bias, wqkv=self.attn.in_proj_weight, bqkv=self.attn.in_proj_bias,
                    wo=self.attn.out_proj.weight, bo=self.attn.out_proj.bias, ln2_w=self.ln_2.weight, ln2_b=self.ln_2.bias,
                    w1=self.mlp.c_fc.weight, b1=self.mlp.c_fc.bias, w2=self.mlp.c_proj.weight, b2=self.mlp.c_proj.bias)

    def forward(self, x):
        """x: [B, N, d] bf16."""
        return HF.transformer_layer(x, self._spec, self._params())


class Transformer(nn.Module):
    def __init__(self, width: int, layers: int, heads: int, attn_mask: torch.Tensor = None):
        super().__init__()
        self.width = width
        self.layers = layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, attn_mask) for _ in range(layers)])

    def forward(self, x):
        return self.resblocks(x)


class VisionTransformer(nn.Module):
    def __init__(self, input_resolution: int, patch_size: int, width: int, layers: int, heads: int, output_dim: int):
        super().__init__()
        self.input_resolution = input_resolution
        self.output_dim = output_dim
        self.patch_size = patch_size
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)  # parameter holder
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))

    def forward(self, x):
        """x: [B, 3, H, W] (fp32 or bf16) -> [B, output_dim] bf16 (reference: model.py:309-335)."""
        x = HF.patch_embed(x, self.conv1.weight, None, self.class_embedding, self.positional_embedding, self.patch_size)
        x = self.ln_pre(x)
        x = self.transformer(x)
        x = self.ln_post(x[:, 0, :].contiguous())
        if self.proj is not None:
            x = HF.linear(x, self.proj, weight_layout="io")
        return x
