"""Projection heads of the M2 encoder on the HIP path (API and parameter names of prj/M2_Encoder/vlmo/modules/heads.py:4-24:
`Pooler.dense.{weight,bias}`, `ITCHead.fc.weight`).

Both heads are one GEMM over the [B, d] class-token features: the projection runs through `antmmf.hip.functional.linear` (bf16 MFMA
GEMM with the bias in the epilogue, wgrad accumulated into the flat gradient arena); `nn.Linear` is only the parameter holder, so
checkpoints map key for key.
"""
import torch
from torch import nn

from antmmf.hip import functional as HF


def _project(holder: nn.Linear, feats: torch.Tensor) -> torch.Tensor:
    """feats @ holder.weight^T (+ bias) on the MI355X; `feats` may be a strided class-token view."""
    return HF.linear(feats.contiguous(), holder.weight, holder.bias)


class ITCHead(nn.Module):
    """Bias-free projection of the class token into the contrastive embedding space."""

    def __init__(self, hidden_size, out_size):
        super().__init__()
        self.fc = nn.Linear(hidden_size, out_size, bias=False)

    def forward(self, x):
        return _project(self.fc, x)


class Pooler(nn.Module):
    """tanh(dense(first token)); the tanh runs in fp32 on the [B, d] result."""

    def __init__(self, hidden_size):
        super().__init__()
        self.dense = nn.Linear(hidden_size, hidden_size)
        self.activation = nn.Tanh()  # kept as an attribute: the reference exposes it

    def forward(self, hidden_states):
        return torch.tanh(_project(self.dense, hidden_states[:, 0]).float())
